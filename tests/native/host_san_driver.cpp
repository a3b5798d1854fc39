// Drives the library's host stages (pt_host.cpp: flatten, camera constants, random tables, blue noise, grid, cell ranges,
// screenshot thresholds) without a GPU, so that they can run under -fsanitize=address,undefined.  Built and run by
// tests/test_host_stages.py.
#include <cstdio>
#include <cstring>
#include <vector>

#include "pt_host.h"

static unsigned g_state = 99u;
static float unit() { g_state = g_state * 1664525u + 1013904223u; return (float)(g_state >> 8) / 16777216.0f; }
static float between(float lo, float hi) { return lo + (hi - lo) * unit(); }

int main() {
	std::vector<pt_object> objs(300);
	for(size_t i = 0; i < objs.size(); i++) {
		pt_object& e = objs[i];
		memset(&e, 0, sizeof e);
		e.type = (int)(i % 4);
		for(int k = 0; k < 3; k++) { e.pos[k] = between(-12, 12); e.dim[k] = between(0.5f, 4); e.color[k] = between(0.2f, 0.9f); }
		e.rot[0] = 1;
		if(i % 2) { e.rot[0] = 0.8f; e.rot[1] = 0.6f; }
		e.refractive_index = i % 5 ? 1.0f : 1.5f;
	}
	const int n = (int)objs.size();
	std::vector<float> rec((size_t)n * PT_F_STRIDE);
	pt::flatten_objects(objs.data(), n, rec.data());
	pt_world w;
	memset(&w, 0, sizeof w);
	w.camera.pos[1] = -30; w.camera.fov = 90; w.camera.plane_width = 10; w.light_rot[1] = 90;
	pt::WorldConsts wc;
	pt::world_consts(w, 321, 123, &wc);
	pt::Pcg rng;
	rng.seed(0, 777);
	std::vector<pt::f3> dirs(PT_TABLE_COUNT);
	std::vector<pt::f2> disc(PT_TABLE_COUNT), samples;
	std::vector<int32_t> offsets;
	pt::make_tables(rng, dirs.data(), disc.data());
	double sum = wc.dim_h + dirs[123].x + disc[456].y;
	for(int spp : { 1, 4, 16, 64, 200 }) for(int gw = 1; gw <= 4; gw++) {
		pt::make_samples(rng, spp, gw, &samples, &offsets);
		sum += (double)samples.size() + offsets.back();
	}
	for(int hint : { 1, 4, 10, 33 }) {
		pt::GridDesc g{};
		std::vector<uint32_t> start, items, ranges;
		if(!pt::build_grid(rec.data(), n, hint, &g, &start, &items)) { fprintf(stderr, "no grid\n"); return 3; }
		pt::object_ranges(rec.data(), n, g, &ranges);
		sum += (double)items.size() + start.back() + ranges.size();
	}
	float thr[255];
	pt::srgb8_thresholds(thr);
	sum += thr[0] + thr[254] + pt::srgb8_encode(0.5f);
	printf("checksum %.6f\n", sum);
	return 0;
}
