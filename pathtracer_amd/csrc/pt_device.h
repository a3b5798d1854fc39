// Device-side building blocks of the path tracer (gfx950).  Everything here is scalar IEEE fp32
// per lane in the reference's operation order: compile with -ffp-contract=off and correctly rounded
// divide/sqrt, never fast-math - primary-hit ids and DDA cell sequences are required to be
// bit-identical to the CPU path (code/raycast.cpp:390-986, 991-1289).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "pt_api.h"
#include "pt_launch.h"
#include "pt_wavescan.h"

#pragma clang fp contract(off)

namespace ptd {

#define PT_DEV __device__ __forceinline__

// Read-only scene data is addressed through the constant address space: a load whose address is
// wave-uniform (the brute-force loop over entities, the per-frame constants) then becomes a scalar
// load into SGPRs instead of 64 identical vector loads; divergent addresses still use vector loads.
#define PT_CONST __attribute__((address_space(4)))
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef const PT_CONST v4f* cf4;
typedef const PT_CONST v2f* cf2;
typedef const PT_CONST uint32_t* cu32;
typedef const PT_CONST int32_t* ci32;

using V3 = ptl::Vec;
using Grid = ptl::GridDev;
typedef const PT_CONST ptl::FrameDev* FrameRef;

struct Scene {
	cf4 geo, dirs;
	cf2 disc, samples;
	cu32 cell_start, cell_items, obj_range, plane_ids;
	cf4 entries;                 // per cell-list entry: bounding sphere, listed-in-neighbour mask, kind, id (ptl::pack_entry)
	ci32 offsets;
	int n, brute, n_planes;
	Grid grid;
};
__device__ __forceinline__ Scene scene_view(const ptl::SceneDev& d) {
	Scene s;
	s.geo = (cf4)d.geo; s.dirs = (cf4)d.dirs;
	s.disc = (cf2)d.disc; s.samples = (cf2)d.samples;
	s.cell_start = (cu32)d.cell_start; s.cell_items = (cu32)d.cell_items; s.obj_range = (cu32)d.obj_range; s.entries = (cf4)d.cell_entries;
	s.offsets = (ci32)d.offsets;
	s.plane_ids = (cu32)d.plane_ids;
	s.n = d.n; s.brute = d.brute; s.n_planes = d.n_planes; s.grid = d.grid;
	return s;
}
PT_DEV V3 v3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
PT_DEV V3 ld3(const PT_CONST V3& v) { return v3(v.x, v.y, v.z); }
PT_DEV V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
PT_DEV V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
PT_DEV V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
PT_DEV V3 operator*(V3 a, float b) { return v3(a.x * b, a.y * b, a.z * b); }
PT_DEV V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
PT_DEV float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
// The fused form, for estimates only: nothing whose bits reach a result may use it.
PT_DEV float fdot(V3 a, V3 b) { return __builtin_fmaf(a.z, b.z, __builtin_fmaf(a.y, b.y, a.x * b.x)); }
PT_DEV V3 cross(V3 a, V3 b) { return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }

// ---- correctly rounded division and square root, the short way ---------------------------------------
// Every quotient and root has to be the IEEE result, bit for bit.  The compiler's a / b is eleven
// instructions: both operands scaled (v_div_scale), v_rcp and one Newton step on the denominator, the
// quotient with two residual corrections (five multiply-adds), the scaling undone (v_div_fmas) and the
// special cases patched (v_div_fixup); its sqrtf is sixteen, seven of them scaling and special cases.
// For operands of ordinary magnitude (non-zero, finite, far from the ends of the exponent range) the
// scaling is by one and the fix-up passes its input through, so the arithmetic in between alone gives
// the same bits; and quotients with a common denominator can then share its reciprocal, which the
// compiler's form cannot (its scaling depends on each numerator).  The functions below run that core
// for every lane and redo the compiler's full sequence for the lanes whose operands are not ordinary
// (a branch the wavefront skips when there are none).
// The long forms are for the rare lane: told so, the compiler lays them out behind the loop, and the short form's path runs
// straight on instead of jumping over them.
#ifdef PT_NO_RARE_HINT
#define PT_RARE(c) (c)
#else
#define PT_RARE(c) __builtin_expect(!!(c), 0)
#endif
#ifdef PT_NO_RARE2
#define PT_RARE2(c) (c)
#define PT_OFTEN2(c) (c)
#else
#define PT_RARE2(c) __builtin_expect(!!(c), 0)
#define PT_OFTEN2(c) __builtin_expect(!!(c), 1)
#endif
constexpr float kPlainLo = 0x1p-40f, kPlainHi = 0x1p40f;      // magnitudes the short forms accept
PT_DEV float min3abs(V3 a) { return __builtin_fminf(__builtin_fminf(__builtin_fabsf(a.x), __builtin_fabsf(a.y)), __builtin_fabsf(a.z)); }
// An upper bound of the three magnitudes that is NaN when any of them is (a three-operand maximum drops
// NaNs, and after a signalling one it can drop a second operand with it).
PT_DEV float sum3abs(V3 a) { return (__builtin_fabsf(a.x) + __builtin_fabsf(a.y)) + __builtin_fabsf(a.z); }
PT_DEV float recip_near(float b) {                   // v_rcp and one Newton step
	const float r = __builtin_amdgcn_rcpf(b);
	const float e = __builtin_fmaf(-b, r, 1.0f);
	return __builtin_fmaf(e, r, r);
}
PT_DEV float quotient(float a, float b, float r) {   // a / b given r = recip_near(b), ordinary operands
	float q = a * r;
	float e = __builtin_fmaf(-b, q, a);
	q = __builtin_fmaf(e, r, q);
	e = __builtin_fmaf(-b, q, a);
	return __builtin_fmaf(e, r, q);
}
PT_DEV float root_core(float x) {                    // sqrt of an ordinary x: v_sqrt, then the neighbour whose residual fits
	const float s = __builtin_amdgcn_sqrtf(x);
	const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
	const float r_down = __builtin_fmaf(-down, s, x), r_up = __builtin_fmaf(-up, s, x);
	float t = r_down <= 0.0f ? down : s;
	t = r_up > 0.0f ? up : t;
	return t;
}
PT_DEV float sq_root(float x) {
	float s = root_core(x);
	if(PT_RARE(!(x >= 0x1p-90f && x <= 0x1p100f))) s = sqrtf(x);
	return s;
}
// Three quotients with one denominator.
PT_DEV V3 operator/(V3 a, float b) {
	const float r = recip_near(b);
	V3 q = v3(quotient(a.x, b, r), quotient(a.y, b, r), quotient(a.z, b, r));
	const float m = __builtin_fabsf(b);
	if(PT_RARE(!(min3abs(a) >= kPlainLo && sum3abs(a) <= kPlainHi && m >= kPlainLo && m <= kPlainHi))) q = v3(a.x / b, a.y / b, a.z / b);
	return q;
}
// 1 / a.x, 1 / a.y, 1 / a.z.  The compiler's own form stays behind a fence for the lanes that need it: its operands are
// a ray's direction, which does not change over a cast, and without the fence the compiler lifts the long form out of its
// guard to run it once for every ray.
PT_DEV V3 recip3(V3 a) {
#ifdef PT_LONG_RECIP
	return v3(1.0f / a.x, 1.0f / a.y, 1.0f / a.z);
#else
	const float rx = recip_near(a.x), ry = recip_near(a.y), rz = recip_near(a.z);
	V3 q = v3(quotient(1.0f, a.x, rx), quotient(1.0f, a.y, ry), quotient(1.0f, a.z, rz));
	if(PT_RARE(!(min3abs(a) >= kPlainLo && sum3abs(a) <= kPlainHi))) {
		V3 b = a;
		asm volatile("" : "+v"(b.x), "+v"(b.y), "+v"(b.z));
		q = v3(1.0f / b.x, 1.0f / b.y, 1.0f / b.z);
	}
	return q;
#endif
}
// a / (float)pi.  The reciprocal of a constant can be given exactly rounded (0x3ea2f983 is the float nearest to 1 / pi);
// a first quotient a * r within one unit of the true one, corrected once by its exact residual, then rounds to the IEEE
// quotient (Markstein's theorem on division with fused multiply-adds), so the second correction of the general form is
// dropped.  Checked against the compiler's division for every float.
PT_DEV V3 over_pi(V3 a) {
	constexpr float pi = 3.14159274101257324f, r = 0x1.45f306p-2f;
	static_assert(r == 0.318309873342514038f, "1 / pi, rounded to nearest");
#ifdef PT_LONG_OVER_PI
	return a / pi;
#else
	V3 q = a * r;
	q = v3(__builtin_fmaf(__builtin_fmaf(-pi, q.x, a.x), r, q.x), __builtin_fmaf(__builtin_fmaf(-pi, q.y, a.y), r, q.y), __builtin_fmaf(__builtin_fmaf(-pi, q.z, a.z), r, q.z));
	if(PT_RARE(!(min3abs(a) >= kPlainLo && sum3abs(a) <= kPlainHi))) q = v3(a.x / pi, a.y / pi, a.z / pi);
	return q;
#endif
}
PT_DEV float length(V3 a) { return sq_root(a.x * a.x + a.y * a.y + a.z * a.z); }
// a / length(a): divides, never multiplies by a reciprocal (code/rt_math.cpp:2166-2168).  Components of at
// least 2^-40 and a sum of squares of at most 2^80 make the root and all three quotients ordinary.
PT_DEV V3 normalise(V3 a) {
	const float ss = a.x * a.x + a.y * a.y + a.z * a.z;
	float l = root_core(ss);
	const float r = recip_near(l);
	V3 q = v3(quotient(a.x, l, r), quotient(a.y, l, r), quotient(a.z, l, r));
	if(PT_RARE(!(min3abs(a) >= kPlainLo && ss <= 0x1p80f))) {
		l = sqrtf(ss);
		q = v3(a.x / l, a.y / l, a.z / l);
	}
	return q;
}
PT_DEV bool is_zero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
// The reference's min/max are ternaries (code/rt_math.cpp:60-74); they differ from fminf/fmaxf on NaN.
PT_DEV float tmin(float a, float b) { return a <= b ? a : b; }
PT_DEV float tmax(float a, float b) { return a >= b ? a : b; }
PT_DEV float clamp_hi(float a, float hi) { return a > hi ? hi : a; }
PT_DEV V3 clamp_hi(V3 a, float hi) { return v3(clamp_hi(a.x, hi), clamp_hi(a.y, hi), clamp_hi(a.z, hi)); }
PT_DEV float blend(float t, float a, float b) {             // lerpCheck, code/rt_math.cpp:222-226
	const float between = a + t * (b - a);
	return t == 0 ? a : (t == 1 ? b : between);
}
PT_DEV V3 mirror(V3 dir, V3 n) {                            // reflectVector, code/rt_math.cpp:2405-2408
	float k = 2 * dot(dir, n);
	return dir - n * k;
}
PT_DEV bool inside_box(V3 p, V3 lo, V3 hi) {                // pointInBox, inclusive (code/rt_math.cpp:3340-3349)
	return p.x >= lo.x && p.x <= hi.x && p.y >= lo.y && p.y <= hi.y && p.z >= lo.z && p.z <= hi.z;
}
PT_DEV V3 rotate(v4f q /* w,x,y,z */, V3 v) {            // Quat*Vec3, code/rt_math.cpp:2902-2907
	V3 u = v3(q.y, q.z, q.w);
	V3 c = cross(u, v);
	V3 t = v3(c.x * 2, c.y * 2, c.z * 2);
	return (v + t * q.x) + cross(u, t);
}

// ---- PCG32 (code/rt_misc.cpp:1152-1251) ----------------------------------------------------------
struct Rng { uint64_t state, inc; };
PT_DEV uint32_t rng_next(Rng& r) {
	uint64_t old = r.state;
	r.state = old * 6364136223846793005ULL + r.inc;
	uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
	uint32_t rot = (uint32_t)(old >> 59u);
	return (xs >> rot) | (xs << ((0u - rot) & 31));
}
PT_DEV void rng_seed(Rng& r, uint64_t initstate, uint64_t initseq) {
	r.state = 0;
	r.inc = (initseq << 1u) | 1u;
	rng_next(r);
	r.state += initstate;
	rng_next(r);
}
template <uint32_t BOUND>
PT_DEV uint32_t rng_bounded(Rng& r) {
	constexpr uint32_t threshold = (0u - BOUND) % BOUND;
	for(;;) {
		uint32_t v = rng_next(r);
		if(PT_OFTEN2(v >= threshold)) return v % BOUND;
	}
}

// ---- scene as the kernels see it ----------------------------------------------------------------
// An entity is one 128-byte record of eight v4f (one cache line, so a lane that tests an entity touches one line, and a
// wave-uniform index turns into wide scalar loads); the slots and their order are ptl::REC_* (pt_launch.h).
enum { G_POS = ptl::REC_POS, G_ROT = ptl::REC_ROT, G_ROTINV = ptl::REC_ROTINV, G_DIM = ptl::REC_DIM, G_SCALE = ptl::REC_SCALE,
       G_SCALEINV = ptl::REC_SCALEINV, M_COLOR = ptl::REC_COLOR, M_EMIT = ptl::REC_EMIT, G_SLOTS = ptl::REC_STRIDE };
// the corners of an unrotated box take the place of the two slots it never reads
enum { G_BOX_LO = ptl::REC_ROTINV, G_BOX_HI = ptl::REC_SCALEINV };
enum { FLAG_ROTATED = 1, FLAG_EVEN = 2 };

// Material of the entity a ray hit: colorLinear|reflectionMod, emitColorLinear|refractiveIndex, roughness.
struct Surface { v4f m0, m1; float rough; };

// ---- analytic intersectors (code/raycast.cpp:390-585) -------------------------------------------
PT_DEV float hit_sphere(V3 lp, V3 ld, V3 sp, float sr, V3& pos, V3& nrm) {
	V3 old = lp;
	lp = lp - sp;
	float d = dot(ld, lp);
	float l = length(lp);
	float b = d * d - l * l + sr * sr;
	if(b < 0) return -1;
	float distance = -(dot(ld, lp));
	float sq = sq_root(b);
	if(distance - sq < 0) {
		if(distance + sq < 0) return -1;
		else distance += sq;
	} else {
		distance -= sq;
	}
	pos = old + ld * distance;
	nrm = normalise(pos - sp);
	return distance;
}

PT_DEV bool touches_sphere(V3 lp, V3 ld, V3 sp, float sr) {  // lineSphereCollision, code/rt_math.cpp:2361-2373
	lp = lp - sp;
	// The quantities the clear-cut decisions are taken from may be fused: their margins are far above any rounding.  The
	// reference's own arithmetic is redone, unfused, only where a decision is close (below).
	float ll = fdot(lp, lp);
	float rr = sr * sr;
	// A ray that starts well inside the sphere touches it, whatever its direction: with |lp|^2 < 0.999 r^2 and a
	// direction no longer than 2, b = d*d - |lp|^2 + r^2 is at least 0.001 r^2 and sqrt(b) exceeds |d| by more than
	// 1e-4 r, four orders of magnitude above the rounding errors of the reference's float arithmetic.  (The bounding
	// sphere of a ground box holds nearly every ray origin of a scene.)  Skipped only when the whole wavefront agrees.
	if(__ballot(!(ll < 0.999f * rr && fdot(ld, ld) <= 4.0f)) == 0) return true;
	float d = fdot(ld, lp);
	// The reference takes two square roots: b = d*d - length(lp)^2 + sr*sr must not be negative, and
	// -d + sqrt(b) must not be negative, which is the case exactly when d <= 0 or sr*sr >= length(lp)^2.
	// Both decisions are clear-cut for almost every ray: when the quantities are further from zero
	// than a margin far above their rounding errors, the answer is known without the roots.
	float margin = 1e-5f * (__builtin_fmaf(d, d, ll) + rr);
	float b_est = __builtin_fmaf(d, d, rr - ll);
	// (plain & and |: every operand is a comparison of values at hand, cheaper to evaluate than to branch around)
	const bool clear_miss = (b_est < -margin) | ((d > 0) & (ll - rr > margin));
	const bool clear_hit = (b_est > margin) & ((d < 0) | (rr - ll > margin));
	bool touches = clear_hit;
	if(PT_RARE(__ballot(!clear_miss && !clear_hit) != 0)) {            // wave-uniform: the roots are skipped only if no lane needs them
		float l = sqrtf(dot(lp, lp));                         // length(lp)
		float dx = dot(ld, lp);
		float b = dx * dx - l * l + sr * sr;
		bool exact = false;
		if(b >= 0) exact = !(-dx + sqrtf(b) < 0);
		if(!clear_miss && !clear_hit) touches = exact;
	}
	return touches;
}

// The slab test of lineBoxIntersection once the box's corners are known.
// (`inv`: the reciprocals of ld's components where the caller has them at hand - a cast that tests several boxes
// with one ray works them out once.)
PT_DEV float hit_slabs(V3 lp, V3 ld, V3 bmin, V3 bmax, V3& pos, V3& nrm, const V3* inv = nullptr) {
	const V3 f = inv ? *inv : recip3(ld);
	const float fx = f.x, fy = f.y, fz = f.z;
	float t1 = (bmin.x - lp.x) * fx;
	float t2 = (bmax.x - lp.x) * fx;
	float t3 = (bmin.y - lp.y) * fy;
	float t4 = (bmax.y - lp.y) * fy;
	float t5 = (bmin.z - lp.z) * fz;
	float t6 = (bmax.z - lp.z) * fz;
	float tn = tmax(tmax(tmin(t1, t2), tmin(t3, t4)), tmin(t5, t6));
	float tf = tmin(tmin(tmax(t1, t2), tmax(t3, t4)), tmax(t5, t6));
	if(tf < 0) return -1;
	if(tn > tf) return -1;
	float distance = tn < 0 ? tf : tn;
	pos = lp + ld * distance;
	// the first of t1 .. t6 that equals the distance names the face (selects, not a chain of branches)
	const bool f1 = distance == t1, f2 = distance == t2, f3 = distance == t3, f4 = distance == t4, f5 = distance == t5, f6 = distance == t6;
	const bool on_x = f1 | f2, on_xy = on_x | f3 | f4;
	nrm.x = f1 ? -1.0f : (f2 ? 1.0f : 0.0f);
	nrm.y = on_x ? 0.0f : (f3 ? -1.0f : (f4 ? 1.0f : 0.0f));
	nrm.z = on_xy ? 0.0f : (f5 ? -1.0f : (f6 ? 1.0f : 0.0f));
	return distance;
}
PT_DEV float hit_box(V3 lp, V3 ld, V3 bpos, V3 bdim, V3& pos, V3& nrm, const V3* inv = nullptr) {
	V3 half = bdim * 0.5f;            // the same floats as bdim / 2
	return hit_slabs(lp, ld, bpos - half, bpos + half, pos, nrm, inv);
}

PT_DEV float hit_cylinder(V3 p, V3 d, V3& pos, V3& nrm) {
	float a = d.x * d.x + d.y * d.y;
	float b = 2 * (p.x * d.x + p.y * d.y);
	float c = (p.x * p.x + p.y * p.y) - 1;
	float disc = b * b - 4 * a * c;
	if(disc < 0) return -1;
	float root = sq_root(disc);
	float t1 = (-b + root) / (2 * a);
	float t2 = (-b - root) / (2 * a);
	float z1 = p.z + d.z * t1;
	float z2 = p.z + d.z * t2;
	int k1 = 0, k2 = 0;   // 0 side, 1 top cap, 2 bottom cap, 3 invalid
	if(z1 > 1) { if(z2 <= 1) { k1 = 1; t1 = (1 - p.z) / d.z; } else k1 = 3; }
	else if(z1 < -1) { if(z2 >= -1) { k1 = 2; t1 = (-1 - p.z) / d.z; } else k1 = 3; }
	if(z2 > 1) { if(z1 <= 1) { k2 = 1; t2 = (1 - p.z) / d.z; } else k2 = 3; }
	else if(z2 < -1) { if(z1 >= -1) { k2 = 2; t2 = (-1 - p.z) / d.z; } else k2 = 3; }
	if(k1 == 3 && k2 == 3) return -1;
	if(t1 < 0 && t2 < 0) return -1;
	float distance = tmin(t1, t2);
	if(distance < 0) distance = tmax(t1, t2);
	pos = p + d * distance;
	int k = distance == t1 ? k1 : k2;
	if(k == 0) {
		if(pos.x == 0 && pos.y == 0) nrm = v3(1, 0, 0);
		else nrm = normalise(v3(pos.x, pos.y, 0));
	}
	else if(k == 1) nrm = v3(0, 0, 1);
	else nrm = v3(0, 0, -1);
	return distance;
}

PT_DEV float hit_cone(V3 p, V3 d, V3& pos, V3& nrm) {
	p.z -= 0.5f;
	float a = d.x * d.x + d.y * d.y - d.z * d.z;
	float b = 2 * p.x * d.x + 2 * p.y * d.y - 2 * p.z * d.z;
	float c = p.x * p.x + p.y * p.y - p.z * p.z;
	float disc = b * b - 4 * a * c;
	if(disc < 0) return -1;
	float root = sq_root(disc);
	float t1 = (-b + root) / (2 * a);
	float t2 = (-b - root) / (2 * a);
	float z1 = p.z + d.z * t1;
	float z2 = p.z + d.z * t2;
	int k1 = 0, k2 = 0;   // 0 side, 2 outside the cone's height
	if(z1 > 0 || z1 < -1) k1 = 2;
	if(z2 > 0 || z2 < -1) k2 = 2;
	if(k1 == 2 && k2 == 2) return -1;
	if(k1 == 2 || k2 == 2) {
		if(k1 == 2) t1 = (-1 - p.z) / d.z;
		else t2 = (-1 - p.z) / d.z;
	}
	if(t1 < 0 && t2 < 0) return -1;
	float distance = tmin(t1, t2);
	if(distance < 0) distance = tmax(t1, t2);
	p.z += 0.5f;
	pos = p + d * distance;
	int k = distance == t1 ? k1 : k2;
	if(k == 0) {
		if(pos.x == 0 && pos.y == 0) {
			nrm = v3(0, 0, 1);
		} else {
			V3 n = normalise(v3(pos.x, pos.y, 0));
			nrm = normalise(v3(n.x, n.y, 1));
		}
	}
	else nrm = v3(0, 0, -1);
	return distance;
}

// linePlaneIntersection (code/rt_math.cpp:2343-2359): the plane through pp with normal pn, either side.
PT_DEV float hit_plane(V3 lp, V3 ld, V3 pp, V3 pn, V3& pos, V3& nrm) {
	const float a = dot(pn, ld);
	if(a == 0) return -1;
	const float distance = dot(pn, pp - lp) / a;
	if(distance >= 0) {
		pos = lp + ld * distance;
		nrm = a < 0 ? pn : -pn;
		return distance;
	}
	return -1;
}

// Entity record loader: one entity = G_SLOTS consecutive float4 (128 bytes, one cache line).
struct GeoRef {
	cf4 base; int n; int i;
	PT_DEV v4f plane(int p) const { return base[(size_t)i * G_SLOTS + (size_t)p]; }
};

// transformToUnitSpace / transformToGlobalSpace (code/raycast.cpp:750-770)
PT_DEV void to_unit(const GeoRef& g, V3 opos, bool rotated, V3 pos, V3 dir, V3& up, V3& ud) {
	v4f si = g.plane(G_SCALEINV);
	V3 sinv = v3(si.x, si.y, si.z);
	pos = pos - opos;
	if(rotated) {
		v4f ri = g.plane(G_ROTINV);
		pos = rotate(ri, pos);
		dir = rotate(ri, dir);
	}
	up = pos * sinv;
	ud = normalise(dir * sinv);
}
PT_DEV void to_world(const GeoRef& g, V3 opos, bool rotated, V3& pos, V3& nrm) {
	v4f s = g.plane(G_SCALE), si = g.plane(G_SCALEINV);
	pos = pos * v3(s.x, s.y, s.z);
	nrm = nrm * v3(si.x, si.y, si.z);
	if(rotated) {
		v4f r = g.plane(G_ROT);
		pos = rotate(r, pos);
		nrm = rotate(r, nrm);
	}
	pos = pos + opos;
	nrm = normalise(nrm);
}

// What a scene contains, as far as kernel variants care: a variant compiled without a feature leaves out
// its code and its registers; the host picks the variant from the scene (SceneDev::features).
enum { FEAT_UNIT_KINDS = 1,     // rotated boxes, ellipsoids, cylinders, cones: the kinds that work in an entity's unit space
       FEAT_GLASS = 2,          // an entity with a refractive index other than 1
       FEAT_ROUGH = 4,          // an entity with roughness (Oren-Nayar)
       FEAT_ALL = 7,            // all of the above: every scene of the path's four primitive kinds
       FEAT_PLANE = 8,          // an infinite plane (GEOM_TYPE_PLANE): tested by every ray that misses the grid, and listed in the
                                // cells of the grid its huge bounding box covers
       FEAT_EVERY = 15 };

// Intersection kinds: the six straight-line variants getIntersection (code/raycast.cpp:813-893)
// dispatches to.  The wave-cooperative cast bins candidates by kind so that a wavefront runs one
// variant at a time.
enum { KIND_SPHERE = 0, KIND_BOX = 1, KIND_BOX_ROTATED = 2, KIND_ELLIPSOID = 3, KIND_CYLINDER = 4, KIND_CONE = 5, KIND_PLANE = 6, KIND_COUNT = 7 };
PT_DEV int kind_of(int type, bool rotated, bool even) {
	if(type == PT_GEOM_SPHERE) return even ? KIND_SPHERE : KIND_ELLIPSOID;
	if(type == PT_GEOM_BOX) return rotated ? KIND_BOX_ROTATED : KIND_BOX;
	if(type == PT_GEOM_PLANE) return KIND_PLANE;
	return type == PT_GEOM_CYLINDER ? KIND_CYLINDER : KIND_CONE;
}
// the plane's normal, rot * (0, 0, 1) (code/raycast.cpp:887): worked out by the host with the reference's operations and kept in
// a slot a plane reads for nothing else
enum { G_PLANE_NORMAL = ptl::REC_ROTINV };

template <int KIND>
PT_DEV float intersect_kind(const GeoRef& g, v4f p0, v4f p3, V3 rp, V3 rd, V3& pos, V3& nrm, const V3* inv = nullptr) {
	V3 opos = v3(p0.x, p0.y, p0.z);
	V3 dim = v3(p3.x, p3.y, p3.z);
	float distance = -1;
	if(KIND == KIND_SPHERE) {
		// the bounding radius of a sphere with three equal dims is dim.x * 0.5f, the radius the reference passes
		distance = hit_sphere(rp, rd, opos, p0.w, pos, nrm);
	} else if(KIND == KIND_BOX) {
		// corners of an unrotated box (pos -/+ dim / 2), kept by the host in two slots such a box never reads
		v4f lo = g.plane(G_BOX_LO), hi = g.plane(G_BOX_HI);
		distance = hit_slabs(rp, rd, v3(lo.x, lo.y, lo.z), v3(hi.x, hi.y, hi.z), pos, nrm, inv);
	} else if(KIND == KIND_PLANE) {
		const v4f pn = g.plane(G_PLANE_NORMAL);
		distance = hit_plane(rp, rd, opos, v3(pn.x, pn.y, pn.z), pos, nrm);
	} else if(KIND == KIND_BOX_ROTATED) {
		v4f ri = g.plane(G_ROTINV);
		V3 lp = rp - opos;
		V3 le = lp + rd;
		lp = rotate(ri, lp);
		le = rotate(ri, le);
		V3 ld = normalise(le - lp);
		distance = hit_box(lp, ld, v3(0, 0, 0), dim, pos, nrm);
		if(distance != -1) {
			v4f r = g.plane(G_ROT);
			pos = rotate(r, pos);
			pos = pos + opos;
			nrm = rotate(r, nrm);
			distance = length(rp - pos);
		}
	} else {
		bool rotated = (__float_as_int(p3.w) >> 8) & FLAG_ROTATED;
		V3 lp, ld;
		to_unit(g, opos, rotated, rp, rd, lp, ld);
		if(KIND == KIND_ELLIPSOID) distance = hit_sphere(lp, ld, v3(0, 0, 0), 0.5f, pos, nrm);
		else if(KIND == KIND_CYLINDER) distance = hit_cylinder(lp, ld, pos, nrm);
		else distance = hit_cone(lp, ld, pos, nrm);
		if(distance != -1) {
			to_world(g, opos, rotated, pos, nrm);
			distance = length(rp - pos);
		}
	}
	return distance;
}

// Cylinder and cone in one body (hit_cylinder, hit_cone): both solve a quadratic in the entity's unit space and differ in
// its coefficients and in what happens at the ends of the shape.  Every lane computes exactly the operations of its own
// shape, in its order; what the two have in common (discriminant, root, the two quotients, the plane crossings, the side
// normal) is executed once for both.
PT_DEV float hit_round(bool cone, V3 p, V3 d, V3& pos, V3& nrm) {
	const float pz = cone ? p.z - 0.5f : p.z;
	const float dd = d.x * d.x + d.y * d.y;
	const float a = cone ? dd - d.z * d.z : dd;
	const float b = cone ? 2 * p.x * d.x + 2 * p.y * d.y - 2 * pz * d.z : 2 * (p.x * d.x + p.y * d.y);
	const float pp = p.x * p.x + p.y * p.y;
	const float c = cone ? pp - pz * pz : pp - 1;
	const float disc = b * b - 4 * a * c;
	if(disc < 0) return -1;
	const float root = sq_root(disc);
	float t1 = (-b + root) / (2 * a);
	float t2 = (-b - root) / (2 * a);
	const float z1 = pz + d.z * t1;
	const float z2 = pz + d.z * t2;
	const float t_low = (-1 - pz) / d.z;      // the plane z = -1: the cylinder's bottom cap, the cone's base
	const float t_high = (1 - pz) / d.z;      // the cylinder's top cap
	int k1 = 0, k2 = 0;                       // what a root stands for: 0 side, 1 top cap, 2 bottom cap or base, 3 nothing
	if(cone) {
		if(z1 > 0 || z1 < -1) k1 = 2;
		if(z2 > 0 || z2 < -1) k2 = 2;
		if(k1 == 2 && k2 == 2) return -1;
		if(k1 == 2) t1 = t_low;
		else if(k2 == 2) t2 = t_low;
	} else {
		if(z1 > 1) { if(z2 <= 1) { k1 = 1; t1 = t_high; } else k1 = 3; }
		else if(z1 < -1) { if(z2 >= -1) { k1 = 2; t1 = t_low; } else k1 = 3; }
		if(z2 > 1) { if(z1 <= 1) { k2 = 1; t2 = t_high; } else k2 = 3; }
		else if(z2 < -1) { if(z1 >= -1) { k2 = 2; t2 = t_low; } else k2 = 3; }
		if(k1 == 3 && k2 == 3) return -1;
	}
	if(t1 < 0 && t2 < 0) return -1;
	float distance = tmin(t1, t2);
	if(distance < 0) distance = tmax(t1, t2);
	pos = v3(p.x, p.y, cone ? pz + 0.5f : p.z) + d * distance;
	const int k = distance == t1 ? k1 : k2;
	if(k == 0) {
		if(pos.x == 0 && pos.y == 0) nrm = cone ? v3(0, 0, 1) : v3(1, 0, 0);
		else {
			const V3 n = normalise(v3(pos.x, pos.y, 0));
			nrm = n;
			if(cone) nrm = normalise(v3(n.x, n.y, 1));
		}
	}
	else nrm = (!cone && k == 1) ? v3(0, 0, 1) : v3(0, 0, -1);
	return distance;
}

// All six kinds in one body, for a wavefront whose lanes hold candidates of different kinds.  The kinds that work in the
// entity's unit space (rotated box, ellipsoid, cylinder, cone) share the transformation in and out; both boxes share the
// slab test, both spheres the sphere test, cylinder and cone the quadratic - with their own operands: a world-space shape
// takes the ray as it is, a unit-space one the transformed ray and the canonical shape (a centre of zero leaves every
// difference and sum as it is: x - 0 and 0 + x are x, 0 - x is the box's lower corner as hit_box computes it).  Every lane
// runs exactly the operation sequence of its intersect_kind variant.
// p1 and p4 are the record's other two early slots, fetched by the caller together with p0 and p3.
constexpr int KIND_MIXED = KIND_COUNT;
template <int FEAT>
PT_DEV float intersect_mixed(int kind, const GeoRef& g, v4f p0, v4f p1, v4f p3, v4f p4, V3 rp, V3 rd, V3& pos, V3& nrm, cu32 range_words, uint32_t& range_lo, uint32_t& range_hi) {
	const V3 opos = v3(p0.x, p0.y, p0.z);
	const V3 dim = v3(p3.x, p3.y, p3.z);
	const bool isbox = kind == KIND_BOX_ROTATED;
	const bool unit = kind >= KIND_BOX_ROTATED && (!(FEAT & FEAT_PLANE) || kind != KIND_PLANE);
	const bool turned = isbox || (unit && ((__float_as_int(p3.w) >> 8) & FLAG_ROTATED));
	V3 a = rp, ld = rd, sinv = v3(1, 1, 1);
	if(unit) {
		sinv = v3(p4.x, p4.y, p4.z);
		a = rp - opos;
		V3 b = isbox ? a + rd : rd;
		if(turned) {
			a = rotate(p1, a);
			b = rotate(p1, b);
		}
		V3 din;
		if(isbox) din = b - a;
		else { a = a * sinv; din = b * sinv; }
		ld = normalise(din);
	}
	float distance;
	if((FEAT & FEAT_PLANE) && kind == KIND_PLANE) {
		distance = hit_plane(rp, rd, opos, v3(p1.x, p1.y, p1.z), pos, nrm);      // p1: G_PLANE_NORMAL
	} else if(kind == KIND_BOX || isbox) {
		V3 bmin, bmax;
		if(isbox) {
			const V3 half = dim * 0.5f;            // the same floats as dim / 2
			bmin = v3(0, 0, 0) - half;
			bmax = v3(0, 0, 0) + half;
		} else {
			bmin = v3(p1.x, p1.y, p1.z);           // G_BOX_LO, G_BOX_HI
			bmax = v3(p4.x, p4.y, p4.z);
		}
		distance = hit_slabs(a, ld, bmin, bmax, pos, nrm);
	} else if(kind == KIND_SPHERE || kind == KIND_ELLIPSOID) {
		const bool even = kind == KIND_SPHERE;
		distance = hit_sphere(a, ld, even ? opos : v3(0, 0, 0), even ? p0.w : 0.5f, pos, nrm);
	} else {
		distance = hit_round(kind == KIND_CONE, a, ld, pos, nrm);
	}
	// anything but a miss: the entity's cell range (for the caller's cell search) is asked for together with the record's late
	// slots (a unit-space distance of zero can still become a positive one in world space)
	if(distance != -1) { range_lo = range_words[2 * g.i]; range_hi = range_words[2 * g.i + 1]; }
	if(unit && distance != -1) {
		if(!isbox) {
			const v4f sc4 = g.plane(G_SCALE);
			pos = pos * v3(sc4.x, sc4.y, sc4.z);
			nrm = nrm * sinv;
		}
		if(turned) {
			const v4f r = g.plane(G_ROT);
			pos = rotate(r, pos);
			nrm = rotate(r, nrm);
		}
		pos = pos + opos;
		if(!isbox) nrm = normalise(nrm);
		distance = length(rp - pos);
	}
	return distance;
}

// getIntersection (code/raycast.cpp:813-893).  p0 = plane G_POS already loaded by the caller.
template <int FEAT = FEAT_ALL>
PT_DEV float intersect(const GeoRef& g, v4f p0, V3 rp, V3 rd, V3& pos, V3& nrm, const V3* inv = nullptr) {
	v4f p3 = g.plane(G_DIM);
	int tf = __float_as_int(p3.w);
	int type = tf & 0xff;
	if(PT_RARE2(type > PT_GEOM_CONE)) {
		if((FEAT & FEAT_PLANE) && type == PT_GEOM_PLANE) return intersect_kind<KIND_PLANE>(g, p0, p3, rp, rd, pos, nrm);
		return -1;
	}
	switch(kind_of(type, (tf >> 8) & FLAG_ROTATED, (tf >> 8) & FLAG_EVEN)) {
		case KIND_SPHERE: return intersect_kind<KIND_SPHERE>(g, p0, p3, rp, rd, pos, nrm);
		case KIND_BOX: return intersect_kind<KIND_BOX>(g, p0, p3, rp, rd, pos, nrm, inv);
		default: break;
	}
	if(!(FEAT & FEAT_UNIT_KINDS)) return -1;          // the scene holds no such entity
	switch(kind_of(type, (tf >> 8) & FLAG_ROTATED, (tf >> 8) & FLAG_EVEN)) {
		case KIND_BOX_ROTATED: return intersect_kind<KIND_BOX_ROTATED>(g, p0, p3, rp, rd, pos, nrm);
		case KIND_ELLIPSOID: return intersect_kind<KIND_ELLIPSOID>(g, p0, p3, rp, rd, pos, nrm);
		case KIND_CYLINDER: return intersect_kind<KIND_CYLINDER>(g, p0, p3, rp, rd, pos, nrm);
		default: return intersect_kind<KIND_CONE>(g, p0, p3, rp, rd, pos, nrm);
	}
}

// ---- grid traversal (code/raycast.cpp:619-741) --------------------------------------------------
struct Walk {
	int cx, cy, cz;          // coord
	int sx, sy, sz;          // step
	float tx, ty, tz;        // tMax
	float dx, dy, dz;        // tDelta
	V3 lo, hi;               // current cell box
};

// Entry test of spatialGridTraverseInit: false when the ray never reaches the grid ("outside").
// On success `p` is the point traversal starts from.
PT_DEV bool grid_entry(const Grid& g, V3 rp, V3 rd, V3& p, const V3* inv = nullptr) {
	V3 bmin = v3((float)(g.cmin[0] * g.cs[0]), (float)(g.cmin[1] * g.cs[1]), (float)(g.cmin[2] * g.cs[2]));
	V3 bmax = v3((float)((g.cmin[0] + g.cnt[0]) * g.cs[0]), (float)((g.cmin[1] + g.cnt[1]) * g.cs[1]), (float)((g.cmin[2] + g.cnt[2]) * g.cs[2]));
	p = rp;
	if(!inside_box(rp, bmin, bmax)) {
		V3 bdim = bmax - bmin;
		V3 bpos = bmin + bdim * 0.5f;     // the same floats as bdim / 2
		V3 ip, in;
		float distance = hit_box(rp, rd, bpos, bdim, ip, in, inv);
		if(distance != -1) p = ip + rd * 0.0001f;
		else return false;
	}
	return true;
}

PT_DEV int imin(int a, int b) { return a < b ? a : b; }
PT_DEV int imax(int a, int b) { return a > b ? a : b; }

#define PT_WALK_AXIS_INIT(A, P, D, CS, CMIN, CNT, C, S, T, DT) \
	C = (int)(P / (float)CS) - CMIN; \
	if(P < 0) C--; \
	C = imin(imax(C, 0), CNT - 1); \
	S = (D == 0) ? 0 : (D > 0 ? 1 : -1); \
	{ int edge = (S == -1 ? C : C + 1) + CMIN; \
	  T = (S != 0) ? ((float)(edge * CS) - P) / D : 3.402823466e+38f; } \
	DT = (float)CS / D; \
	if(DT < 0) DT = -DT;

PT_DEV void walk_start(const Grid& g, V3 p, V3 d, Walk& w) {
	PT_WALK_AXIS_INIT(x, p.x, d.x, g.cs[0], g.cmin[0], g.cnt[0], w.cx, w.sx, w.tx, w.dx)
	PT_WALK_AXIS_INIT(y, p.y, d.y, g.cs[1], g.cmin[1], g.cnt[1], w.cy, w.sy, w.ty, w.dy)
	PT_WALK_AXIS_INIT(z, p.z, d.z, g.cs[2], g.cmin[2], g.cnt[2], w.cz, w.sz, w.tz, w.dz)
	w.lo = v3((float)((w.cx + g.cmin[0]) * g.cs[0]), (float)((w.cy + g.cmin[1]) * g.cs[1]), (float)((w.cz + g.cmin[2]) * g.cs[2]));
	w.hi = v3((float)((w.cx + 1 + g.cmin[0]) * g.cs[0]), (float)((w.cy + 1 + g.cmin[1]) * g.cs[1]), (float)((w.cz + 1 + g.cmin[2]) * g.cs[2]));
}

#define PT_WALK_STEP(C, S, T, DT, CNT, CMIN, CS, LO, HI) \
	{ C += S; \
	  if(C < 0 || C >= CNT) return false; \
	  T += DT; \
	  if(S < 0) { HI = LO; LO = (float)((C + CMIN) * CS); } \
	  else { LO = HI; HI = (float)((C + CMIN + 1) * CS); } }

// spatialGridTraverseNext: the axis with the smallest tMax advances; ties go x<y ? (x<z ? X : Z) : (y<z ? Y : Z).
PT_DEV bool walk_next(const Grid& g, Walk& w) {
	if(w.tx < w.ty) {
		if(w.tx < w.tz) PT_WALK_STEP(w.cx, w.sx, w.tx, w.dx, g.cnt[0], g.cmin[0], g.cs[0], w.lo.x, w.hi.x)
		else PT_WALK_STEP(w.cz, w.sz, w.tz, w.dz, g.cnt[2], g.cmin[2], g.cs[2], w.lo.z, w.hi.z)
	} else {
		if(w.ty < w.tz) PT_WALK_STEP(w.cy, w.sy, w.ty, w.dy, g.cnt[1], g.cmin[1], g.cs[1], w.lo.y, w.hi.y)
		else PT_WALK_STEP(w.cz, w.sz, w.tz, w.dz, g.cnt[2], g.cmin[2], g.cs[2], w.lo.z, w.hi.z)
	}
	return true;
}

// The walker without the cell box: the cooperative cast re-derives a cell's box from its
// coordinates, which gives the same floats (both are (coord + coordMin [+ 1]) * cellSize converted).
struct WalkLite { int cx, cy, cz, sx, sy, sz; float tx, ty, tz, dx, dy, dz; };
PT_DEV WalkLite lite_of(const Walk& w) {
	WalkLite l;
	l.cx = w.cx; l.cy = w.cy; l.cz = w.cz; l.sx = w.sx; l.sy = w.sy; l.sz = w.sz;
	l.tx = w.tx; l.ty = w.ty; l.tz = w.tz; l.dx = w.dx; l.dy = w.dy; l.dz = w.dz;
	return l;
}
#define PT_LITE_STEP(C, S, T, DT, CNT) { C += S; if(C < 0 || C >= CNT) return false; T += DT; }
PT_DEV bool lite_next(const Grid& g, WalkLite& w) {
	if(w.tx < w.ty) {
		if(w.tx < w.tz) PT_LITE_STEP(w.cx, w.sx, w.tx, w.dx, g.cnt[0])
		else PT_LITE_STEP(w.cz, w.sz, w.tz, w.dz, g.cnt[2])
	} else {
		if(w.ty < w.tz) PT_LITE_STEP(w.cy, w.sy, w.ty, w.dy, g.cnt[1])
		else PT_LITE_STEP(w.cz, w.sz, w.tz, w.dz, g.cnt[2])
	}
	return true;
}

#define PT_HASH_INIT 1469598103934665603ULL
PT_DEV uint64_t hash_cell(uint64_t h, int index) { return (h ^ (uint64_t)(uint32_t)(index + 1)) * 1099511628211ULL; }

PT_DEV Surface surface_of(const Scene& sc, int index) {
	Surface m;
	m.m0 = sc.geo[(size_t)index * G_SLOTS + M_COLOR];
	m.m1 = sc.geo[(size_t)index * G_SLOTS + M_EMIT];
	m.rough = sc.geo[(size_t)index * G_SLOTS + G_SCALE].w;
	return m;
}

// castRayAll (code/raycast.cpp:895-926): every entity, bounding-sphere prefilter first.  The
// traverse-init the reference runs before it (code/raycast.cpp:1101) only matters for `outside`.
// The loop index is wave-uniform, so the entity records arrive through scalar loads, and the
// surface of the winning entity is latched on the way instead of being gathered afterwards.
// GRID_GATE false is the editor's pick (code/app.cpp:1637, 3236): castRayAll with `outside` false, no grid involved.
template <int FEAT = FEAT_ALL, bool GRID_GATE = true>
PT_DEV int cast_all(const Scene& sc, V3 rp, V3 rd, V3& hpos, V3& hnrm, Surface& surf) {
	V3 entry;
	const V3 inv = recip3(rd);          // for the grid's box and every upright box of the scene
	bool outside = false;               // the ray never meets the grid: only planes are tested (code/raycast.cpp:903)
	if(GRID_GATE && !grid_entry(sc.grid, rp, rd, entry, &inv)) {
		if(!(FEAT & FEAT_PLANE)) return 0;      // and this scene has none
		outside = true;
	}
	int id = 0;
	float best = 3.402823466e+38f;
	for(int i = 0; i < sc.n; i++) {
		GeoRef g = { sc.geo, sc.n, i };
		v4f p0 = g.plane(G_POS);
		bool plane = false;
		if(FEAT & FEAT_PLANE) {
			plane = (__float_as_int(g.plane(G_DIM).w) & 0xff) == PT_GEOM_PLANE;      // wave-uniform
			if(outside && !plane) continue;
		}
		if(plane || touches_sphere(rp, rd, v3(p0.x, p0.y, p0.z), p0.w)) {      // a plane has no bounding sphere (code/raycast.cpp:909)
			V3 p, q;
			float distance = intersect<FEAT>(g, p0, rp, rd, p, q, &inv);
			if(distance > 0 && distance < best) {
				best = distance;
				id = i + 1;
				hpos = p;
				hnrm = q;
				surf = surface_of(sc, i);
			}
		}
	}
	return id;
}

// The rays that never meet the grid test the scene's planes, all of them, and the last one hit is the result, whatever its
// distance, a distance of zero included (code/raycast.cpp:937-955).
PT_DEV int cast_planes(const Scene& sc, V3 rp, V3 rd, V3& hpos, V3& hnrm) {
	int id = 0;
	for(int k = 0; k < sc.n_planes; k++) {
		const int oi = (int)sc.plane_ids[k];
		GeoRef g = { sc.geo, sc.n, oi };
		const v4f p0 = g.plane(G_POS);
		V3 p, q;
		const float distance = intersect_kind<KIND_PLANE>(g, p0, p0, rp, rd, p, q);
		if(distance != -1) { id = oi + 1; hpos = p; hnrm = q; }
	}
	return id;
}

// castRay (code/raycast.cpp:928-986): walk the grid, accept a hit only inside the current cell,
// stop at the first cell that produced one.  Items are in ascending id, so on equal distance the
// lowest id wins, as in the reference.
template <bool HASH, int FEAT = FEAT_EVERY>
PT_DEV int cast_grid(const Scene& sc, V3 rp, V3 rd, V3& hpos, V3& hnrm, Surface& surf, uint64_t& hash) {
	V3 start;
	if(!grid_entry(sc.grid, rp, rd, start)) {
		int outside_id = 0;
		if(FEAT & FEAT_PLANE) {
			outside_id = cast_planes(sc, rp, rd, hpos, hnrm);
			if(outside_id != 0) surf = surface_of(sc, outside_id - 1);
		}
		return outside_id;
	}
	Walk w;
	walk_start(sc.grid, start, rd, w);
	int id = 0;
	float best = 3.402823466e+38f;
	do {
		if(HASH) hash = hash_cell(hash, w.cz * sc.grid.cnt[1] * sc.grid.cnt[0] + w.cy * sc.grid.cnt[0] + w.cx);     // the reference's cell index
		const int cell = w.cx * sc.grid.stride[0] + w.cy * sc.grid.stride[1] + w.cz * sc.grid.stride[2];           // where the cell sits here (ptl::grid_layout)
		uint32_t k0 = sc.cell_start[cell] & ptl::CELL_START_MASK, k1 = sc.cell_start[cell + 1] & ptl::CELL_START_MASK;
		for(uint32_t k = k0; k < k1; k++) {
			int oi = (int)sc.cell_items[k];
			GeoRef g = { sc.geo, sc.n, oi };
			v4f p0 = g.plane(G_POS);
			V3 p, q;
			float distance = intersect<FEAT>(g, p0, rp, rd, p, q);
			if(distance > 0 && distance < best && inside_box(p, w.lo, w.hi)) {
				best = distance;
				id = oi + 1;
				hpos = p;
				hnrm = q;
			}
		}
		if(id != 0) break;
	} while(walk_next(sc.grid, w));
	if(id != 0) surf = surface_of(sc, id - 1);
	return id;
}

// The cast of the small passes (tile-cost probes, debug outputs, ray queries, the cast histogram): every feature compiled in.
template <bool HASH>
PT_DEV int cast(const Scene& sc, V3 rp, V3 rd, V3& hpos, V3& hnrm, Surface& surf, uint64_t& hash) {
	if(sc.brute) return cast_all<FEAT_EVERY>(sc, rp, rd, hpos, hnrm, surf);
	return cast_grid<HASH, FEAT_EVERY>(sc, rp, rd, hpos, hnrm, surf, hash);
}

// ---- wave-cooperative grid cast ------------------------------------------------------------------
// Same results as cast_grid, organised for wave64.  A segment of the cast has three phases:
//   walk     every walking lane steps COOP_CELLS cells ahead and only notes, per non-empty cell, where
//            its list starts and how long it is (one descriptor in LDS);
//   expand   the lists of all lanes are flattened into one sequence of (ray, cell, entry) items that
//            the 64 lanes share evenly, whatever the ray they belong to: each item is one 32-byte
//            read, a duplicate check against the ray's previous cell and a conservative
//            bounding-sphere filter; survivors go into a pool, tagged with their intersection kind;
//   test     the pool is grouped by kind and the expensive analytic intersections run one kind at a
//            time with lanes taken from the bins, so a wavefront executes one straight-line variant
//            with many lanes busy instead of the union of all variants with a few.
// Collection runs ahead of the decision which cell wins; the reference's rule (first visited cell
// with a hit inside it; nearest hit in that cell; lowest id on equal distance;
// code/raycast.cpp:957-983) is restored exactly by a 64-bit key (cell ordinal | distance bits | id)
// reduced with an LDS atomic min per ray.
#ifndef PT_COOP_CELLS
#define PT_COOP_CELLS 8              // (with lists that repeat the previous cell's skipped, eight cells and 274 descriptors beat seven and 306 by 0.5 - 1.3 %)
#endif
#ifndef PT_COOP_MIN_WALKING
#define PT_COOP_MIN_WALKING 40
#endif
constexpr int COOP_CELLS = PT_COOP_CELLS;               // cells a lane records per segment
#ifndef PT_COOP_POOL
#define PT_COOP_POOL 256
#endif
#ifndef PT_COOP_DESCS
#define PT_COOP_DESCS 274
#endif
constexpr int COOP_POOL = PT_COOP_POOL;                 // candidates awaiting their intersection test
constexpr int COOP_DESCS = PT_COOP_DESCS;               // non-empty cells the wavefront can note per segment
constexpr int COOP_MIN_WALKING = PT_COOP_MIN_WALKING;   // fewer walking lanes than this: return and refill
#ifndef PT_COOP_ITEMS
#define PT_COOP_ITEMS 2
#endif
constexpr int COOP_ITEMS = PT_COOP_ITEMS;                 // list entries a lane expands per trip
#ifndef PT_COOP_HEAD_WORDS
#define PT_COOP_HEAD_WORDS 32
#endif
constexpr int COOP_HEAD_WORDS = PT_COOP_HEAD_WORDS;       // words of the list-head bitmap (at most 64: one lane scans one word)
constexpr uint32_t COOP_SEQ_MAX = 32u * COOP_HEAD_WORDS;  // positions a list may start at in one segment
static_assert(PT_COOP_CELLS <= 8, "the cell ordinal has three bits in a descriptor");
#ifndef PT_COOP_MIX_ALL
#define PT_COOP_MIX_ALL 128
#endif
#ifndef PT_COOP_MIX_UNIT
#define PT_COOP_MIX_UNIT 128
#endif
constexpr int COOP_MIX_ALL = PT_COOP_MIX_ALL;     // at most this many candidates in all: chunks of mixed kinds
constexpr int COOP_MIX_UNIT = PT_COOP_MIX_UNIT;   // at most this many unit-space candidates: mixed chunks for them
constexpr uint64_t COOP_NONE = ~0ull;

struct CoopShared {                           // one per wavefront
	v2f ray[64][3];                           // origin, direction of each lane's ray (x y | z dx | dy dz)
	uint32_t cells[COOP_CELLS][64];           // packed coordinates of the cells recorded this segment
	uint8_t ncell[64];
	unsigned long long best[64];              // winning key per ray
	v2f res[64][3];                           // hit position, normal of the winner, packed the same way
	uint32_t pool[COOP_POOL];                 // candidates: owner lane << 26 | kind << 23 | entity
	uint32_t sorted[COOP_POOL];               // the same, grouped by kind
	uint32_t count[KIND_COUNT + 2];           // per kind, [KIND_COUNT] = total
	uint32_t fill[KIND_COUNT + 2];
	uint32_t desc_k0[COOP_DESCS];             // first entry of the cell's list
	uint32_t turn;                            // segments this wavefront has run: which lane is served first
	uint32_t head[COOP_HEAD_WORDS];           // bit p set: a list starts at position p of the flattened sequence
	uint16_t rank0[COOP_HEAD_WORDS];          // lists that start before the word's first position
	uint32_t desc_at[COOP_DESCS];             // position of that entry in the flattened sequence << 12 |
	                                          // owner lane | cell ordinal << 6 | face the ray entered through << 9 (7: none)
#ifdef PT_COOP_TIMING
	unsigned long long tacc[16];              // per-phase cycle sums of this wavefront
#endif
};
#if !defined(PT_COOP_TIMING) && !defined(PT_TRACE_WAVES_PER_SIMD)
static_assert(sizeof(CoopShared) <= 10240, "four wavefronts per SIMD need the per-wave LDS share within 10 KB");
#endif

// Diagnostic builds: -DPT_COOP_STATS counts events of the cooperative cast, -DPT_COOP_TIMING adds up
// the cycles each wavefront spends per phase (build one or the other: the event atomics distort timing).
#if defined(PT_COOP_STATS) || defined(PT_COOP_TIMING)
__device__ unsigned long long g_coop_stats[16];
#define PT_COUNT_(i, v) atomicAdd((unsigned long long*)&g_coop_stats[i], (unsigned long long)(v))
#endif
#ifdef PT_COOP_STATS
#define PT_STAT(i, v) PT_COUNT_(i, v)
#else
#define PT_STAT(i, v)
#endif
#ifdef PT_COOP_TIMING
#define PT_TICK(var) long long var = clock64()
#define PT_TOCK(sh_, i, var) { long long now_ = clock64(); if(lane == 0) (sh_).tacc[i] += (unsigned long long)(now_ - var); var = now_; }
// -DPT_COOP_TIMING_WAITS: slots 6 .. 9 hold, instead of the chunk counts, the cycles up to the issue of the walk's list-bound
// loads (6), the wait for them (7), the wait for the expand step's list entries (8) and for a chunk's first record loads (9);
// the waits are forced by s_waitcnt, so the phases' own totals shift a little
#ifdef PT_COOP_TIMING_WAITS
#define PT_TOCK_ISSUED(sh_, i, var) PT_TOCK(sh_, i, var)
#define PT_TOCK_LANDED(sh_, i, var) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); PT_TOCK(sh_, i, var) }
#else
#define PT_TOCK_ISSUED(sh_, i, var)
#define PT_TOCK_LANDED(sh_, i, var)
#endif
#else
#define PT_TICK(var)
#define PT_TOCK(sh_, i, var)
#define PT_TOCK_ISSUED(sh_, i, var)
#define PT_TOCK_LANDED(sh_, i, var)
#endif
// -DPT_ASM_MARKS: named comments in the assembly output, so that a script can count the instructions between them
#ifdef PT_ASM_MARKS
#define PT_MARK(name) asm volatile("; PTMARK " name ::: "memory")
#else
#define PT_MARK(name)
#endif
PT_DEV void wave_sync() {
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
PT_DEV int lanes_under(uint64_t mask) {   // set bits of mask in lanes below this one
	return (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
}
PT_DEV uint32_t uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
PT_DEV uint32_t pack_cell(int x, int y, int z) { return (uint32_t)x | ((uint32_t)y << 10) | ((uint32_t)z << 20); }
PT_DEV bool range_has(uint32_t lo, uint32_t hi, uint32_t c) {
	uint32_t cx = c & 1023, cy = (c >> 10) & 1023, cz = (c >> 20) & 1023;
	return cx >= (lo & 1023) && cx <= (hi & 1023) && cy >= ((lo >> 10) & 1023) && cy <= ((hi >> 10) & 1023) &&
	       cz >= ((lo >> 20) & 1023) && cz <= ((hi >> 20) & 1023);
}
PT_DEV void put6(v2f (&slot)[3], V3 a, V3 b) { slot[0] = (v2f){ a.x, a.y }; slot[1] = (v2f){ a.z, b.x }; slot[2] = (v2f){ b.y, b.z }; }
PT_DEV void get6(const v2f (&slot)[3], V3& a, V3& b) {
	v2f u = slot[0], v = slot[1], w = slot[2];
	a = v3(u.x, u.y, v.x); b = v3(v.y, w.x, w.y);
}
// Conservative reject: true when the ray's line misses the entity's (slightly enlarged) bounding
// sphere or the sphere lies wholly behind the origin.  No exact intersector can report a hit then,
// so skipping the entity does not change any result.  No divide, no square root.
// This test decides nothing but which entities get the exact test, and its margins are a hundred times the rounding of
// its arithmetic: it is the one place that uses fused multiply-adds (a third fewer instructions).
// line_only: the sphere may lie behind the origin.  For a cone in the cone-exact variants: the reference's cone intersector,
// given a ray whose line grazes the apex of a cone behind it, can find one of two nearly equal roots above the apex and the other
// below, replaces the first by the crossing of the base plane - which lies ahead - and reports that (code/raycast.cpp:352-388).
PT_DEV bool sphere_rejects(V3 rp, V3 rd, v4f p0, bool line_only = false) {
	V3 l = rp - v3(p0.x, p0.y, p0.z);
	float dd = fdot(rd, rd), dl = fdot(rd, l), ll = fdot(l, l);
	float r = __builtin_fmaf(p0.w, 1.002f, 1e-3f);
	float r2 = r * r;
	float dl2 = dl * dl;
	float slack = 1e-5f * __builtin_fmaf(dd, ll, dl2);
	float disc = __builtin_fmaf(-dd, ll - r2, dl2);
	bool may_hit = (disc + slack >= 0) && (line_only || dl <= 0 || ll <= __builtin_fmaf(r2, 1.0001f, slack));
	return !may_hit;
}

// One chunk of up to 64 candidates of one kind (or of any, KIND_MIXED), one per lane, taken from list[base .. end).  Called by
// the whole wavefront: the rendezvous between the atomic min and the winner's write-back must be reached by all 64 lanes.
template <int KIND, int FEAT>
PT_DEV void coop_chunk(const Scene& sc, CoopShared& sh, const uint32_t* list, uint32_t base, uint32_t end, int lane) {
PT_MARK("chunk_begin");
	uint32_t i = base + (uint32_t)lane;
	unsigned long long key = COOP_NONE;
	int owner = 0;
	if(lane == 0) PT_STAT(2, 1);
	if(i < end) PT_STAT(3, 1);
	V3 pos = v3(0, 0, 0), nrm = v3(0, 0, 0);
	PT_TICK(tc);
	float distance = -1;
	int id = 0, mixed_kind = 0;
	V3 rp = v3(0, 0, 0), rd = v3(0, 0, 1);
	if(i < end) {
		uint32_t entry = list[i];
		owner = (int)(entry >> 26);
		id = (int)(entry & 0x7fffffu);
		mixed_kind = (int)((entry >> 23) & 7u);
		get6(sh.ray[owner], rp, rd);
	}
	PT_TOCK(sh, 2, tc);
PT_MARK("chunk_fetched");
	uint32_t lo = 0, hi = 0;
	if(i < end) {
		GeoRef g = { sc.geo, sc.n, id };
		v4f p0 = g.plane(G_POS), p3 = g.plane(G_DIM);
		// a chunk of mixed kinds asks for the whole first half of the record at once (G_ROTINV and G_SCALEINV, or an upright
		// box's corners, in p1 and p4): one memory round trip instead of two for every unit-space candidate
		v4f p1 = p0, p4 = p0;
		if(KIND == KIND_MIXED) { p1 = g.plane(G_ROTINV); p4 = g.plane(G_SCALEINV); }
		PT_TOCK_LANDED(sh, 9, tc);
		distance = KIND == KIND_MIXED ? intersect_mixed<FEAT>(mixed_kind, g, p0, p1, p3, p4, rp, rd, pos, nrm, sc.obj_range, lo, hi) : intersect_kind<KIND>(g, p0, p3, rp, rd, pos, nrm);
	}
	PT_TOCK(sh, 3, tc);
PT_MARK("chunk_intersected");
	// The hit counts for the first recorded cell that lists the entity and whose box holds the hit point, bounds
	// included (code/raycast.cpp:957-983, pointInBox).  Cell bounds are integers (coordinate * cell size), exact as
	// floats, so per axis the cells whose closed box holds the point are floor(p / size) and, when p sits on its lower
	// bound, the one before.  Nearly every point sits on no bound: then one cell can hold it, and the search through
	// the recorded cells is a comparison of packed coordinates.
	const bool hit = i < end && distance > 0;
	int ordinal = -1;
	int q[3] = { 0, 0, 0 };
	uint32_t on_bound = 0;
	if(hit) {
		// the entity's cell range: fetched for hits only (two in five candidates); a chunk of mixed kinds has asked for it already
		if(KIND != KIND_MIXED) { lo = sc.obj_range[2 * id]; hi = sc.obj_range[2 * id + 1]; }
		const float pc[3] = { pos.x, pos.y, pos.z };
		bool listed = true;
		int rel[3];
		#pragma unroll
		for(int a = 0; a < 3; a++) {
			// in floats throughout: a cell bound is coordinate * size, and the product of two integers held exactly rounds like the integer
			// product converted - the same bound either way, without the integer multiplies (two ordinary instructions each here)
			const float cs = (float)sc.grid.cs[a];
			float kf = floorf(pc[a] * sc.grid.inv_cs[a]);                  // within one of the true quotient's floor
			const float below = kf * cs, above = (kf + 1.0f) * cs;
			const bool down = below > pc[a], up = !down && above <= pc[a];
			kf = down ? kf - 1.0f : up ? kf + 1.0f : kf;
			const float bound = down ? kf * cs : up ? above : below;
			if(bound == pc[a]) on_bound |= 1u << a;
			const int k = (int)kf;
			q[a] = k;
			rel[a] = k - sc.grid.cmin[a];
			const int lo_a = (int)((lo >> (10 * a)) & 1023u), hi_a = (int)((hi >> (10 * a)) & 1023u);
			listed = listed && (uint32_t)(rel[a] - lo_a) <= (uint32_t)(hi_a - lo_a);
		}
		const uint32_t target = (on_bound == 0 && listed) ? pack_cell(rel[0], rel[1], rel[2]) : 0xffffffffu;    // no recorded cell looks like the latter
		#pragma unroll
		for(int o = COOP_CELLS - 1; o >= 0; o--) {
			if(sh.cells[o][owner] == target) ordinal = o;
		}
		if(ordinal >= (int)sh.ncell[owner]) ordinal = -1;           // a leftover of an earlier segment; any true match lies before it
	}
	if(PT_RARE2(__ballot(hit && on_bound != 0) != 0)) {
		// a point on a cell's lower bound: the interval of admissible coordinates per axis, cut to the entity's range
		if(hit && on_bound != 0) {
			int amin[3], amax[3];
			#pragma unroll
			for(int a = 0; a < 3; a++) {
				const int cmin = sc.grid.cmin[a];
				const int first = ((on_bound >> a) & 1u) ? q[a] - 1 : q[a];
				amin[a] = imax(first - cmin, (int)((lo >> (10 * a)) & 1023u));
				amax[a] = imin(q[a] - cmin, (int)((hi >> (10 * a)) & 1023u));
			}
			const int n = (int)sh.ncell[owner];
			#pragma unroll
			for(int o = COOP_CELLS - 1; o >= 0; o--) {
				const uint32_t c = sh.cells[o][owner];
				const int cx = (int)(c & 1023u), cy = (int)((c >> 10) & 1023u), cz = (int)((c >> 20) & 1023u);
				if(o < n && cx >= amin[0] && cx <= amax[0] && cy >= amin[1] && cy <= amax[1] && cz >= amin[2] && cz <= amax[2]) ordinal = o;
			}
		}
	}
PT_MARK("chunk_searched");
	if(ordinal >= 0) {
		key = ((unsigned long long)ordinal << 55) | ((unsigned long long)__float_as_uint(distance) << 23) | (unsigned long long)id;
		atomicMin(&sh.best[owner], key);
		PT_STAT(8, 1);
	}
	PT_TOCK(sh, 4, tc);
	wave_sync();
PT_MARK("chunk_synced");
	if(key != COOP_NONE && sh.best[owner] == key) put6(sh.res[owner], pos, nrm);
	PT_TOCK(sh, 5, tc);
PT_MARK("chunk_end");
#if defined(PT_COOP_TIMING) && !defined(PT_COOP_TIMING_WAITS)
	if(lane == 0) sh.tacc[KIND == KIND_MIXED ? 9 : 6 + (KIND >= 3 ? 1 : 0) + (KIND >= 4 ? 1 : 0)] += 1;     // chunks: direct kinds, ellipsoid, cylinder/cone, mixed
#endif
}

// Tests everything in the pool and empties it.  Whole wavefront.
template <int FEAT>
PT_DEV void coop_drain(const Scene& sc, CoopShared& sh, int lane) {
PT_MARK("drain_begin");
	const uint32_t total = uniform(sh.count[KIND_COUNT]);
	if(total == 0) return;
#ifdef PT_COOP_STATS
	if(lane == 0) {
		if(total > 64u && total <= 80u) PT_STAT(14, 1);                     // a second chunk for at most 16 candidates
		if(total > 128u && total <= 144u) PT_STAT(14, 1);
		if(total <= 32u) PT_STAT(15, 1);                                    // one chunk less than half full
	}
#endif
	if(total <= 64u) {
		// one chunk of mixed kinds, whatever their order: straight from the pool
		coop_chunk<KIND_MIXED, FEAT>(sc, sh, sh.pool, 0, total, lane);
	} else {
		uint32_t off[KIND_COUNT + 1];
		off[0] = 0;
		#pragma unroll
		for(int k = 0; k < KIND_COUNT; k++) off[k + 1] = off[k] + ((FEAT & FEAT_PLANE) || k != KIND_PLANE ? uniform(sh.count[k]) : 0u);
		for(uint32_t i = (uint32_t)lane; i < total; i += 64) {
			uint32_t e = sh.pool[i];
			uint32_t kind = (e >> 23) & 7u;
			uint32_t base = kind == 0 ? off[0] : kind == 1 ? off[1] : kind == 2 ? off[2] : kind == 3 ? off[3] : kind == 4 ? off[4] : (!(FEAT & FEAT_PLANE) || kind == 5) ? off[5] : off[6];
			sh.sorted[base + atomicAdd(&sh.fill[kind], 1u)] = e;
		}
		wave_sync();
		// Few candidates: chunks of mixed kinds cost less than a nearly empty chunk per kind.
		if(total <= (uint32_t)COOP_MIX_ALL) {
			for(uint32_t base = 0; base < total; base += 64) coop_chunk<KIND_MIXED, FEAT>(sc, sh, sh.sorted, base, total, lane);
		} else {
			for(uint32_t base = off[0]; base < off[1]; base += 64) coop_chunk<KIND_SPHERE, FEAT>(sc, sh, sh.sorted, base, off[1], lane);
			for(uint32_t base = off[1]; base < off[2]; base += 64) coop_chunk<KIND_BOX, FEAT>(sc, sh, sh.sorted, base, off[2], lane);
			if(off[6] - off[2] <= (uint32_t)COOP_MIX_UNIT) {
				for(uint32_t base = off[2]; base < off[6]; base += 64) coop_chunk<KIND_MIXED, FEAT>(sc, sh, sh.sorted, base, off[6], lane);
			} else {
				for(uint32_t base = off[2]; base < off[3]; base += 64) coop_chunk<KIND_BOX_ROTATED, FEAT>(sc, sh, sh.sorted, base, off[3], lane);
				for(uint32_t base = off[3]; base < off[4]; base += 64) coop_chunk<KIND_ELLIPSOID, FEAT>(sc, sh, sh.sorted, base, off[4], lane);
				for(uint32_t base = off[4]; base < off[5]; base += 64) coop_chunk<KIND_CYLINDER, FEAT>(sc, sh, sh.sorted, base, off[5], lane);
				for(uint32_t base = off[5]; base < off[6]; base += 64) coop_chunk<KIND_CONE, FEAT>(sc, sh, sh.sorted, base, off[6], lane);
			}
			if(FEAT & FEAT_PLANE) {
				for(uint32_t base = off[6]; base < off[7]; base += 64) coop_chunk<KIND_PLANE, FEAT>(sc, sh, sh.sorted, base, off[7], lane);
			}
		}
	}
PT_MARK("drain_tail");
	if(lane < KIND_COUNT + 2) { sh.count[lane] = 0; sh.fill[lane] = 0; }
	wave_sync();
}

// What a lane keeps of a cast between calls: the walker's cell, next-crossing parameters and their
// increments.  Step signs come back from the ray's direction.
struct CoopLane {
	int cx, cy, cz;
	float tx, ty, tz, dx, dy, dz;
	bool walking;
};
PT_DEV int sign_of(float d) { return d == 0 ? 0 : (d > 0 ? 1 : -1); }
PT_DEV void coop_lane_reset(CoopLane& st) { st.walking = false; }

// All 64 lanes must call this together; `active` lanes own a ray, the others only lend a hand.
// A lane with `fresh` set starts the cast of (rp, rd); an active lane without it continues the cast
// it was in.  The call returns as soon as fewer than COOP_MIN_WALKING lanes are still walking, so
// the caller can shade the finished rays and hand their lanes new ones instead of idling them until
// the longest walk of the wavefront ends; `done` tells which lanes have their result.
// CONES: also the reference cone intersector's phantom hits, which the bounding-sphere filter otherwise drops (in the expand step).
template <int FEAT, bool CONES = false>
PT_DEV int cast_grid_coop(const Scene& sc, CoopShared& sh, int lane, bool active, bool fresh, CoopLane& st, V3 rp, V3 rd, V3& hpos,
                          V3& hnrm, Surface& surf, bool& done) {
	int result = 0;
	if(active && fresh) {
		coop_lane_reset(st);
		V3 start;
		if(grid_entry(sc.grid, rp, rd, start)) {
			Walk full;
			walk_start(sc.grid, start, rd, full);
			st.cx = full.cx; st.cy = full.cy; st.cz = full.cz;
			st.tx = full.tx; st.ty = full.ty; st.tz = full.tz;
			st.dx = full.dx; st.dy = full.dy; st.dz = full.dz;
			st.walking = true;
		} else if(FEAT & FEAT_PLANE) {
			result = cast_planes(sc, rp, rd, hpos, hnrm);      // the ray never meets the grid: the planes, and this lane is done
		}
		put6(sh.ray[lane], rp, rd);
	}
	bool walking = active && st.walking;

	if(lane == 0 && __ballot(walking) != 0) PT_STAT(9, 1);
	while(__ballot(walking) != 0) {
#ifdef PT_COOP_TIMING
		if(lane == 0) sh.tacc[1] += 1;
#endif
		if(lane == 0) PT_STAT(1, 1);
		PT_TICK(t0);
PT_MARK("seg_begin");
		// ---- reset the wavefront's bins --------------------------------------------------------
		if(lane < KIND_COUNT + 2) { sh.count[lane] = 0; sh.fill[lane] = 0; }
		if(lane < COOP_HEAD_WORDS) sh.head[lane] = 0;
		if(lane == 0) sh.turn = sh.turn + 1u;
		sh.best[lane] = COOP_NONE;
		sh.ncell[lane] = 0;
		wave_sync();

PT_MARK("walk_begin");
		// ---- walk: note the non-empty cells of the next COOP_CELLS steps ---------------------------
		// A probe copy of the walker runs through the whole segment first, so the list bounds of
		// all its cells are fetched in one memory round trip.
		// The steps are written without branches: which axis advances is three comparisons, and everything that
		// follows from it a select (the ties of spatialGridTraverseNext: x<y ? (x<z ? X : Z) : (y<z ? Y : Z)).
		int m = 0;                            // cells of this segment that lie inside the grid
		bool more = walking;                  // the probe stands on a cell of the grid
		int cx = st.cx, cy = st.cy, cz = st.cz;
		float tx = st.tx, ty = st.ty, tz = st.tz;
		uint32_t cc[COOP_CELLS], ka[COOP_CELLS], kb[COOP_CELLS], fc[COOP_CELLS];
		#pragma unroll
		for(int j = 0; j < COOP_CELLS; j++) { cc[j] = 0; ka[j] = 0; kb[j] = 0; fc[j] = 7; }
		if(walking) {
			V3 from, dir;
			get6(sh.ray[lane], from, dir);
			const int sx = sign_of(dir.x), sy = sign_of(dir.y), sz = sign_of(dir.z);
			// the face a step enters the next cell through: -x +x -y +y -z +z, 7 for the cell the segment starts in
			const uint32_t fx = (uint32_t)sx >> 31, fy = 2u + ((uint32_t)sy >> 31), fz = 4u + ((uint32_t)sz >> 31);
			const int nx = sc.grid.cnt[0], ny = sc.grid.cnt[1], nz = sc.grid.cnt[2];
			// the cell's index follows the steps: integer multiplies cost two ordinary instructions each
			const int s0 = sc.grid.stride[0], s1 = sc.grid.stride[1], s2 = sc.grid.stride[2];
			int at = cx * s0 + cy * s1 + cz * s2;
			const int ix = sx > 0 ? s0 : sx < 0 ? -s0 : 0, iy = sy > 0 ? s1 : sy < 0 ? -s1 : 0, iz = sz > 0 ? s2 : sz < 0 ? -s2 : 0;
			uint32_t face = 7;
			int pc[COOP_CELLS];
			#pragma unroll
			for(int j = 0; j < COOP_CELLS; j++) {
				pc[j] = more ? at : 0;
				cc[j] = pack_cell(cx, cy, cz);
				fc[j] = face;
				m += more ? 1 : 0;
				const bool xy = tx < ty, xz = tx < tz, yz = ty < tz;
				const bool ax = xy && xz, ay = !xy && yz;
				const bool az = !(ax || ay);
				face = ax ? fx : ay ? fy : fz;
				cx += ax ? sx : 0; cy += ay ? sy : 0; cz += az ? sz : 0;
				at += ax ? ix : ay ? iy : iz;
				const float ux = tx + st.dx, uy = ty + st.dy, uz = tz + st.dz;
				tx = ax ? ux : tx; ty = ay ? uy : ty; tz = az ? uz : tz;
				more = more && (uint32_t)cx < (uint32_t)nx && (uint32_t)cy < (uint32_t)ny && (uint32_t)cz < (uint32_t)nz;
			}
			// (a cell's two bounds as one eight-byte request instead of two of four: measured, no difference on either grid frame)
			#pragma unroll
			for(int j = 0; j < COOP_CELLS; j++) { ka[j] = sc.cell_start[pc[j]]; kb[j] = sc.cell_start[pc[j] + 1]; }
			// A cell whose every entity is also listed in the cell the ray comes from holds nothing new: its list is skipped (the
			// flags ride above the list's first index, ptl::CELL_FLAG_SHIFT; the segment's first cell has no face and no such bit).
			#pragma unroll
			for(int j = 0; j < COOP_CELLS; j++) {
#ifndef PT_COOP_NO_CELL_SKIP
				const bool repeat = (((ka[j] >> ptl::CELL_FLAG_SHIFT) >> fc[j]) & 1u) != 0u;
#else
				const bool repeat = false;
#endif
				ka[j] &= ptl::CELL_START_MASK;
				kb[j] = repeat ? ka[j] : kb[j] & ptl::CELL_START_MASK;
			}
		}
		PT_TOCK_ISSUED(sh, 6, t0);
		PT_TOCK_LANDED(sh, 7, t0);
PT_MARK("walk_stepped");
		// Descriptors and sequence positions are dealt ray by ray: two wave prefix sums (lists and entries per lane, in
		// registers) give every lane the slot and the position its first list starts at, its other lists follow.  The
		// lanes are served in an order that starts one lane later every segment, so no lane is last for ever.  A list is
		// taken while a descriptor is free and its first entry lies inside the bitmap; slots and positions grow together,
		// so everything after the first refusal is refused too (a list may run past the bitmap's end: it is then the last).
		const uint32_t rot = uniform(sh.turn) & 63u;
		uint32_t my_lists = 0, my_entries = 0;
		#pragma unroll
		for(int j = 0; j < COOP_CELLS; j++) {
			const uint32_t c = j < m ? kb[j] - ka[j] : 0u;
			my_lists += c > 0 ? 1u : 0u;
			my_entries += c;
		}
		uint32_t slot, start, all_lists, all_entries;
		{
			const uint32_t upto_l = wave_inclusive_sum(my_lists), upto_e = wave_inclusive_sum(my_entries);
			all_lists = (uint32_t)__builtin_amdgcn_readlane((int)upto_l, 63);
			all_entries = (uint32_t)__builtin_amdgcn_readlane((int)upto_e, 63);
			const uint32_t skipped_l = rot ? (uint32_t)__builtin_amdgcn_readlane((int)upto_l, (int)rot - 1) : 0u;   // of the lanes served last
			const uint32_t skipped_e = rot ? (uint32_t)__builtin_amdgcn_readlane((int)upto_e, (int)rot - 1) : 0u;
			const bool early = (uint32_t)lane >= rot;
			slot = (upto_l - my_lists) + (early ? 0u - skipped_l : all_lists - skipped_l);
			start = (upto_e - my_entries) + (early ? 0u - skipped_e : all_entries - skipped_e);
		}
PT_MARK("walk_dealt");
		// A lane's lists follow one another from its slot and position.  The first list that finds no descriptor free or
		// would start past the bitmap is turned away, and with it everything behind it (slot and position stay where they
		// are, so what follows meets the same refusal); its cell and the cells after it wait for the next segment.
		int turned_at = COOP_CELLS;
		#pragma unroll
		for(int j = 0; j < COOP_CELLS; j++) {
			const uint32_t c = j < m ? kb[j] - ka[j] : 0u;
			const bool fits = slot < (uint32_t)COOP_DESCS && start < COOP_SEQ_MAX;
			const bool take = c > 0 && fits;
			turned_at = (c > 0 && !fits) ? imin(turned_at, j) : turned_at;
			if(take) {
				sh.desc_k0[slot] = ka[j];
				sh.desc_at[slot] = (start << 12) | (uint32_t)lane | ((uint32_t)j << 6) | (fc[j] << 9);
				atomicOr(&sh.head[start >> 5], 1u << (start & 31u));       // start < COOP_SEQ_MAX for every list taken
			}
			slot += take ? 1u : 0u;
			start += take ? c : 0u;
			sh.cells[j][lane] = cc[j];             // all of them: what lies past the recorded ones is never looked at
		}
		const bool turned_away = turned_at < COOP_CELLS;
		const int recorded = imin(m, turned_at);
#ifdef PT_COOP_STATS
		if(walking) {
			PT_STAT(10, 1);
			if(my_lists == 0 && more) PT_STAT(11, 1);                       // nothing to collect in the whole segment, and the ray goes on
			if(my_lists == 0 && !more) PT_STAT(12, 1);                      // nothing to collect, and the ray leaves the grid
			if(my_lists == 1) PT_STAT(13, 1);
		}
#endif
PT_MARK("walk_placed");
		// the first refusal in serving order ends the sequence: what was placed before it is contiguous from zero
		uint32_t ndesc = all_lists, total = all_entries;
		{
			const uint64_t refusals = __ballot(turned_away);
			if(refusals != 0) {
				const uint64_t in_order = rot ? (refusals >> rot) | (refusals << (64u - rot)) : refusals;
				const int first = (int)(((uint32_t)__ffsll((unsigned long long)in_order) - 1u + rot) & 63u);
				ndesc = (uint32_t)__builtin_amdgcn_readlane((int)slot, first);
				total = (uint32_t)__builtin_amdgcn_readlane((int)start, first);
			}
		}
		if(walking) {
			sh.ncell[lane] = (uint8_t)recorded;
			PT_STAT(4, recorded);
		}
		wave_sync();
		PT_TOCK(sh, 10, t0);

PT_MARK("expand_begin");
		// ---- expand: the lists of all lanes as one sequence, 64 entries per round ------------------
		// Which list an entry position belongs to: descriptors are in position order, so it is the number of
		// list heads at or before the position, less one - a prefix count per bitmap word plus a popcount.
		{
			const uint32_t mine = lane < COOP_HEAD_WORDS ? (uint32_t)__popc(sh.head[lane < COOP_HEAD_WORDS ? lane : 0]) : 0u;
			const uint32_t upto = wave_inclusive_sum(mine);
			if(lane < COOP_HEAD_WORDS) sh.rank0[lane] = (uint16_t)(upto - mine);
		}
		wave_sync();
PT_MARK("expand_loop");
		for(uint32_t base = 0; base < total; base += 64 * COOP_ITEMS) {
PT_MARK("trip_begin");
			// COOP_ITEMS independent items per lane and trip, so their LDS and memory round trips overlap
			uint32_t item[COOP_ITEMS], d[COOP_ITEMS];
			#pragma unroll
			for(int u = 0; u < COOP_ITEMS; u++) {
				item[u] = base + (uint32_t)(u * 64 + lane);
				// positions past the bitmap's end belong to the last list; positions past the sequence's end are not used
				const bool past = item[u] >= COOP_SEQ_MAX;
				const uint32_t pos = (item[u] < total && !past) ? item[u] : 0u;
				const uint32_t heads = sh.head[pos >> 5] & (0xffffffffu >> (31u - (pos & 31u)));
				d[u] = (past && item[u] < total) ? ndesc - 1u : (uint32_t)sh.rank0[pos >> 5] + (uint32_t)__popc(heads) - 1u;
			}
PT_MARK("trip_ranked");
			uint32_t who[COOP_ITEMS];
			v4f e[COOP_ITEMS];
			bool live[COOP_ITEMS];
			V3 rop[COOP_ITEMS], rod[COOP_ITEMS];
			#pragma unroll
			for(int u = 0; u < COOP_ITEMS; u++) {
				live[u] = item[u] < total;
				who[u] = sh.desc_at[d[u]];
				const uint32_t k = live[u] ? sh.desc_k0[d[u]] + (item[u] - (who[u] >> 12)) : 0u;
				e[u] = sc.entries[k];
				get6(sh.ray[who[u] & 63u], rop[u], rod[u]);       // the owner's ray: read while the entry travels
			}
			PT_TOCK_ISSUED(sh, 11, t0);
			PT_TOCK_LANDED(sh, 8, t0);
PT_MARK("trip_loaded");
			bool cone_need[COOP_ITEMS];               // CONES: the item is a cone that the sphere test turned away
			uint32_t cone_id[COOP_ITEMS];
			#pragma unroll
			for(int u = 0; u < COOP_ITEMS; u++) { cone_need[u] = false; cone_id[u] = 0; }
			#pragma unroll
			for(int u = 0; u < COOP_ITEMS; u++) {
				if(live[u]) {
					const int owner = (int)(who[u] & 63u);
					const uint32_t tag = __float_as_uint(e[u].w);
					PT_STAT(5, 1);
					// an entity that is also listed in the cell the ray came from was collected there
#ifndef PT_COOP_NO_DUP
					const bool dup = (((tag >> 3) & 63u) >> ((who[u] >> 9) & 7u)) & 1u;        // face 7 (no previous cell) reads a zero bit
#else
					const bool dup = false;
#endif
					if(dup) PT_STAT(6, 1);
					if(!dup) {
						const V3 op = rop[u], od = rod[u];
						v4f ball = e[u];                 // the id's bits move the centre by less than the radius was enlarged by
						ball.w = __uint_as_float(tag & 0xffff0000u);
#ifndef PT_COOP_NO_FILTER
						const bool keep = !sphere_rejects(op, od, ball, CONES && (tag & 7u) == (uint32_t)KIND_CONE);
						// CONES: a cone the ray stays clear of is looked at once more, below
						if(CONES && !keep && (tag & 7u) == (uint32_t)KIND_CONE) {
							cone_need[u] = true;
							cone_id[u] = ptl::entry_id(__float_as_uint(e[u].x), __float_as_uint(e[u].y), __float_as_uint(e[u].z), tag);
						}
						if(keep) {
#else
						if(true) {
#endif
							uint32_t kind = tag & 7u;
							const uint32_t id = ptl::entry_id(__float_as_uint(e[u].x), __float_as_uint(e[u].y), __float_as_uint(e[u].z), tag);
							uint32_t slot = atomicAdd(&sh.count[KIND_COUNT], 1u);
							atomicAdd(&sh.count[kind], 1u);
							sh.pool[slot] = ((uint32_t)owner << 26) | (kind << 23) | id;
						} else PT_STAT(7, 1);
					}
				}
			}
			// CONES: a cone the ray stays clear of is still a candidate when the ray runs parallel to one of its generators - the
			// reference's intersector then divides by a rounded zero and reports a hit where the ray crosses the base plane
			// (pt::cone_slopes has the formula and the two float4 behind the entity records that this reads).  Both items' words
			// are asked for before either is looked at: one wait per trip.
			if constexpr (CONES) {
				bool any = false;
				#pragma unroll
				for(int u = 0; u < COOP_ITEMS; u++) any = any || cone_need[u];
				if(any) {
					v4f s0[COOP_ITEMS], s1[COOP_ITEMS];
					#pragma unroll
					for(int u = 0; u < COOP_ITEMS; u++) {
						cf4 slopes = sc.geo + ((size_t)sc.n * G_SLOTS + 2 * (size_t)cone_id[u]);      // (entity 0's when the item needs none: any valid address)
						s0[u] = slopes[0];
						s1[u] = slopes[1];
					}
					#pragma unroll
					for(int u = 0; u < COOP_ITEMS; u++) {
						const V3 od = rod[u];
						const float ax = fdot(od, v3(s0[u].x, s0[u].y, s0[u].z)), az = fdot(od, v3(s1[u].x, s1[u].y, s1[u].z));
						const float px = ax * ax, pz = az * az;
						const float a = __builtin_fmaf(s0[u].w, px, __builtin_fmaf(s1[u].w, fdot(od, od) - pz, -pz));
						if(cone_need[u] && !(__builtin_fabsf(a) >= 0.02f * __builtin_fmaf(2.0f, pz, a))) {
							uint32_t slot = atomicAdd(&sh.count[KIND_COUNT], 1u);
							atomicAdd(&sh.count[KIND_CONE], 1u);
							sh.pool[slot] = ((who[u] & 63u) << 26) | ((uint32_t)KIND_CONE << 23) | cone_id[u];
						}
					}
				}
			}
PT_MARK("trip_filtered");
			wave_sync();
			if(uniform(sh.count[KIND_COUNT]) > (uint32_t)(COOP_POOL - 64 * COOP_ITEMS)) {
				PT_TOCK(sh, 11, t0);
				coop_drain<FEAT>(sc, sh, lane);
				PT_TOCK(sh, 12, t0);
			}
		}
		PT_TOCK(sh, 11, t0);

PT_MARK("expand_end");
		// ---- test what is left in the pool ------------------------------------------------------------
		coop_drain<FEAT>(sc, sh, lane);
		PT_TOCK(sh, 12, t0);

PT_MARK("verdict_begin");
		// ---- verdict: every list of every recorded cell has been tested ------------------------------
		if(walking) {
			unsigned long long key = recorded > 0 ? sh.best[lane] : COOP_NONE;
			if(key != COOP_NONE) {
				result = (int)(key & 0x7fffffu) + 1;
				get6(sh.res[lane], hpos, hnrm);
				walking = false;
			} else if(PT_OFTEN2(recorded == m)) {
				if(!more) walking = false;   // the ray has left the grid; else the probe stands on the first cell of the next segment
			} else {
				// stopped early: resume at the first cell not recorded
				WalkLite probe;
				V3 from, dir;
				get6(sh.ray[lane], from, dir);
				probe.sx = sign_of(dir.x); probe.sy = sign_of(dir.y); probe.sz = sign_of(dir.z);
				probe.dx = st.dx; probe.dy = st.dy; probe.dz = st.dz;
				probe.cx = st.cx; probe.cy = st.cy; probe.cz = st.cz; probe.tx = st.tx; probe.ty = st.ty; probe.tz = st.tz;
				for(int j = 0; j < recorded; j++) lite_next(sc.grid, probe);
				cx = probe.cx; cy = probe.cy; cz = probe.cz; tx = probe.tx; ty = probe.ty; tz = probe.tz;
			}
			st.cx = cx; st.cy = cy; st.cz = cz; st.tx = tx; st.ty = ty; st.tz = tz;
		}
		PT_TOCK(sh, 13, t0);
PT_MARK("verdict_end");
		if(__popcll(__ballot(walking)) < COOP_MIN_WALKING) break;
	}
PT_MARK("cast_tail");
	st.walking = walking;
	done = active && !walking;
	if(result != 0) surf = surface_of(sc, result - 1);
	return result;
}

// ---- optics (code/raycast.cpp:194-242) -----------------------------------------------------------
// clamp(-1, 1, x) in the reference binds as clamp(value=-1, lo=1, hi=x) and evaluates to (1 > x ? x : 1).
PT_DEV float cap_one(float x) { return 1.0f > x ? x : 1.0f; }
PT_DEV V3 refract_dir(V3 incident, V3 normal, float ior) {
	float cosi = cap_one(dot(incident, normal));
	float etai = 1, etat = ior;
	V3 n = normal;
	if(cosi < 0) cosi = -cosi;
	else { float t = etai; etai = etat; etat = t; n = -normal; }
	float eta = etai / etat;
	float k = 1 - eta * eta * (1 - cosi * cosi);
	if(k < 0) return v3(0, 0, 0);
	return incident * eta + n * (eta * cosi - sq_root(k));
}
PT_DEV float fresnel(V3 incident, V3 normal, float ior) {
	float cosi = cap_one(dot(incident, normal));
	float etai = 1, etat = ior;
	if(cosi > 0) { float t = etai; etai = etat; etat = t; }
	float sint = etai / etat * sq_root(tmax(0.0f, 1 - cosi * cosi));
	if(sint >= 1) return 1;
	float cost = sq_root(tmax(0.0f, 1 - sint * sint));
	cosi = fabsf(cosi);
	float rs = ((etat * cosi) - (etai * cost)) / ((etat * cosi) + (etai * cost));
	float rp = ((etai * cosi) - (etat * cost)) / ((etai * cosi) + (etat * cost));
	return (rs * rs + rp * rp) / 2;
}

// ---- one bounce of shading (code/raycast.cpp:1109-1260) -----------------------------------------
// Returns true when the path continues.  rp/rd are updated to the next ray.
template <int FEAT = FEAT_ALL>
PT_DEV bool shade(const Scene& sc, FrameRef fr, int id, const Surface& surf, V3 hp, V3 hn, int bounce, V3& rp, V3& rd, V3& att, V3& col, Rng& rng) {
	if(id != 0) {
		v4f m0 = surf.m0, m1 = surf.m1;
		V3 albedo = v3(m0.x, m0.y, m0.z), emit = v3(m1.x, m1.y, m1.z);
		float refl = m0.w, ior = m1.w;
		if(PT_RARE2(!is_zero(emit))) col = col + att * emit;
		if((FEAT & FEAT_GLASS) && ior != 1.0f) {
			float ratio = fresnel(rd, hn, ior);
			V3 through = refract_dir(rd, hn, ior);
			if(is_zero(through)) ratio = 1;
			V3 bias = hn * 0.0001f;
			bool outside = dot(rd, hn) < 0;
			float u = (float)(int)rng_bounded<1000>(rng) * 0.001f;   // randomFloatPCG(0, 1, 0.001f): {0..999}/1000
			if(u < ratio) {
				rd = mirror(rd, hn);
				rp = hp + (outside ? bias : -bias);
			} else {
				rd = through;
				rp = hp + (outside ? -bias : bias);
			}
			att = att * albedo;
		} else {
			rp = hp + hn * 0.0001f;
			V3 spec = v3(0, 0, 0), diff = v3(0, 0, 0);
			if(refl != 0.0f) spec = mirror(rd, hn);
			if(refl != 1.0f) {
				v4f t = sc.dirs[rng_bounded<20000>(rng)];
				diff = v3(t.x, t.y, t.z);
				if(dot(diff, hn) < 0) diff = mirror(diff, hn);
			}
			V3 out = -rd;
			V3 in = v3(blend(refl, diff.x, spec.x), blend(refl, diff.y, spec.y), blend(refl, diff.z, spec.z));
			rd = in;                                     // not normalised when 0 < refl < 1, as in the reference
			float cosine = blend(1 - refl, 1, dot(in, hn));
			att = over_pi(att * albedo) * cosine;
			float rough = surf.rough;
			if((FEAT & FEAT_ROUGH) && rough != 0.0f) {                          // Oren-Nayar (code/raycast.cpp:1199-1221)
				float r2 = rough * rough;
				float a = 1 - 0.5f * (r2 / (r2 + 0.33f));
				float b = 0.45f * (r2 / (r2 + 0.09f));
				float cos_in = dot(rd, hn);
				float cos_out = dot(out, hn);
				V3 lp = normalise(rd - hn * cos_in);
				V3 vp = normalise(out - hn * cos_out);
				float cos_phi = dot(lp, vp);
				float th_in = acosf(cos_in);
				float th_out = acosf(cos_out);
				float alpha = tmax(th_in, th_out);
				float beta = tmin(th_in, th_out);
				float orna = a + (b * tmax(0.0f, cos_phi) * sinf(alpha) * tanf(beta));
				att = att * orna;
			}
		}
		return !is_zero(att);
	} else if(bounce == 0) {
		col = col + ld3(fr->bg);
		return false;
	} else {
		float ld = dot(rd, ld3(fr->light_dir));
		ld = ld < 0 ? 0.0f : ld;
		col = col + att * (ld3(fr->ambient) + ld3(fr->light) * ld);
		return false;
	}
}

// ---- camera (code/raycast.cpp:1040-1084) --------------------------------------------------------
PT_DEV V3 pixel_origin(FrameRef fr, int x, int y) {
	float px = (float)x / (float)fr->width, py = (float)y / (float)fr->height;
	return (ld3(fr->top_left) + ld3(fr->right_width) * px) + ld3(fr->down_height) * py;
}
PT_DEV void camera_ray(const Scene& sc, FrameRef fr, V3 origin, v2f s, int lens_index, V3& rp, V3& rd) {
	rp = (origin + ld3(fr->right_px) * s.x) + ld3(fr->down_px) * s.y;
	V3 cam = ld3(fr->cam_pos);
	float aperture = fr->aperture;
	if(aperture == 0.0f) {
		rd = normalise(rp - cam);
	} else {
		V3 focal = v3(0, 0, 0);
		V3 back = ld3(fr->back);
		V3 through = rp - cam;
		V3 plane_pos = cam + ld3(fr->cam_dir) * fr->focal_distance;
		float a = dot(back, through);                        // linePlaneIntersection, code/rt_math.cpp:2343-2359
		if(a != 0) {
			float distance = dot(back, (plane_pos - rp)) / a;
			if(distance >= 0) focal = rp + through * distance;
		}
		v2f lens = sc.disc[lens_index % 20000];
		float ox = lens.x * aperture;
		float oy = lens.y * aperture;
		rp = rp + (ld3(fr->right) * ox + ld3(fr->down) * oy);
		rd = normalise(focal - rp);
	}
}

// Stream ids of the per-sample seed policy: one stream per pixel (lens table start) and one per
// camera sample; `stride` bounds the samples a tile may hold.
#define PT_STREAM_STRIDE 4096ULL
PT_DEV int lens_start(FrameRef fr, int x, int y) {
	Rng r;
	rng_seed(r, fr->seed, (uint64_t)y * (uint64_t)fr->width + (uint64_t)x);
	return (int)rng_bounded<20000>(r);
}
PT_DEV void sample_stream(FrameRef fr, int x, int y, int si, uint64_t pass, Rng& r) {
	uint64_t wh = (uint64_t)fr->width * (uint64_t)fr->height;
	uint64_t pix = (uint64_t)y * (uint64_t)fr->width + (uint64_t)x;
	rng_seed(r, fr->seed + pass, wh + pix * PT_STREAM_STRIDE + (uint64_t)si);
}

}  // namespace ptd
