// Host-side stages of the render core: everything the reference does on the main thread at render
// start (code/app.cpp:1161-1357) before the tile jobs run.  Scalar IEEE fp32 in the reference's
// operation order; build with -ffp-contract=off.
#pragma once
#include <cstdint>
#include <vector>

#include "pt_api.h"

namespace pt {

struct f3 { float x, y, z; };
struct f2 { float x, y; };
struct q4 { float w, x, y, z; };

// Derived per-frame constants (camera basis, image plane, light, linear colours).
struct WorldConsts {
	f3 cam_pos, cam_dir, cam_up, cam_right, top_left;
	float dim_w, dim_h, near_dist;
	f3 light_dir, bg, ambient, light;
	float focal_distance, aperture;
};

struct GridDesc {
	int32_t coord_min[3], cell_count[3], cell_size[3];
	int64_t cells() const { return (int64_t)cell_count[0] * cell_count[1] * cell_count[2]; }
};

// PCG32 stream (code/rt_misc.cpp:1152-1251).
struct Pcg {
	uint64_t state, inc;
	void seed(uint64_t initstate, uint64_t initseq);
	uint32_t next();
	uint32_t bounded(uint32_t bound);
	int range(int from, int to);                       // randomIntPCG
	float quantised(float from, float to, float prec); // randomFloatPCG
};

void flatten_objects(const pt_object* objs, int n, float* records);
void world_consts(const pt_world& w, int width, int height, WorldConsts* out);
void make_tables(Pcg& rng, f3* dirs, f2* disc);
// Blue-noise samples bucketed per pixel tile; offsets has grid_width^2+1 entries.
void make_samples(Pcg& rng, int spp_wanted, int grid_width, std::vector<f2>* samples, std::vector<int32_t>* offsets);
// Grid dimensions alone (spatialGridInit, code/raycast.cpp:595-617).
bool grid_desc(const float* records, int n, int hint, GridDesc* g);
// Returns false when the scene has no finite extent or the grid would exceed 2^28 cells.
bool build_grid(const float* records, int n, int hint, GridDesc* g, std::vector<uint32_t>* cell_start,
                std::vector<uint32_t>* cell_items);

// Per entity two words: its inclusive cell range (10 bits per axis, low word = lower corner) and its
// intersection kind (3 bits split over the two spare bit pairs).  False when a range does not fit.
bool object_ranges(const float* records, int n, const GridDesc& g, std::vector<uint32_t>* ranges);

// A plane's normal rot * (0, 0, 1) (code/raycast.cpp:887) from its flattened record.
void plane_normal(const float* record, float out[3]);
// Radius of a sphere about the entity's position that encloses it (the cooperative cast's conservative filter); infinite for a plane.
float filter_radius(const float* record);
// The same with a centre of its own: a cone's smallest enclosing sphere sits below the entity's position, towards the base.
float filter_sphere(const float* record, float centre[3]);
// What tells whether a ray's direction is parallel, or nearly, to one of a cone's generators: two of the cone's axes in the world
// and two ratios of its squared extents (eight floats; pt_host.cpp has the formula).
void cone_slopes(const float* record, float out[8]);

// Screenshot encoding of one channel (floatToSrgb then colorFloatToInt, code/rt_math.cpp:3634-3640,351-353, as
// code/app.cpp:1544 combines them) and the 255 smallest inputs in [0, 1] at which its output steps up to
// 1 .. 255, found by bisection over float bit patterns (the expression is monotone).
uint8_t srgb8_encode(float linear);
void srgb8_thresholds(float* out255);

}  // namespace pt
