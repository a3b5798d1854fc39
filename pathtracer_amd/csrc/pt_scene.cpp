// Scene files in the reference application's own format (saveScene / loadScene,
// code/raycast.cpp:1578-1604): the raw image of one `World` struct followed by the raw images of its
// `Object` array, as laid out by an x64 compiler (MSVC and g++ agree on these structs):
//
//   World  224 bytes: camera{pos 0, rot 12, dim 24, nearDist 32, farDist 36, fov 40, ovecs 44..80},
//                     objects{data* 80, count 88, reserved 92, startSize 96}, backgroundColor 104,
//                     ambientLightColor 116, globalLightDir 128, globalLightColor 140,
//                     focalPointDistance 152, apertureSize 156, lockFocalPoint 160, focalPoint 164,
//                     globalLightRot 176, precalculated linear colours 184..220
//   Object 172 bytes: id 0, pos 4, rot(w,x,y,z) 16, dim 32, color 44, geometry{type 56, radius 60,
//                     box 64}, material{emitColor 76, reflectionMod 88, refractiveIndex 92,
//                     roughness 96, emitLinear 100}, markedForDeletion 112, precalc 116..172
//
// Only the authoring members are read; everything precalculated (and the stale pointer at offset
// 80) is ignored, as the application recomputes it at render start.
#include <cstdio>
#include <cstring>
#include <vector>

#include "pt_api.h"

namespace {
constexpr size_t kWorldBytes = 224, kObjectBytes = 172;
template <typename T> T rd(const unsigned char* p, size_t at) { T v; memcpy(&v, p + at, sizeof v); return v; }
template <typename T> void wr(unsigned char* p, size_t at, T v) { memcpy(p + at, &v, sizeof v); }
}

extern "C" {

int pt_scene_load(const char* path, pt_world* world, pt_object* objects, int capacity, int* count) {
	if(!path || !world || !count) return PT_ERR_ARG;
	FILE* f = fopen(path, "rb");
	if(!f) return PT_ERR_ARG;
	unsigned char w[kWorldBytes];
	if(fread(w, 1, kWorldBytes, f) != kWorldBytes) { fclose(f); return PT_ERR_ARG; }
	int n = rd<int32_t>(w, 88);
	if(n < 0 || n > (1 << 24)) { fclose(f); return PT_ERR_LIMIT; }
	memset(world, 0, sizeof *world);
	for(int k = 0; k < 3; k++) {
		world->camera.pos[k] = rd<float>(w, 0 + 4 * k);
		world->camera.rot[k] = rd<float>(w, 12 + 4 * k);
		world->background_color[k] = rd<float>(w, 104 + 4 * k);
		world->ambient_color[k] = rd<float>(w, 116 + 4 * k);
		world->light_color[k] = rd<float>(w, 140 + 4 * k);
	}
	world->camera.plane_width = rd<float>(w, 24);
	world->camera.fov = rd<float>(w, 40);
	world->focal_distance = rd<float>(w, 152);
	world->aperture = rd<float>(w, 156);
	world->light_rot[0] = rd<int32_t>(w, 176);
	world->light_rot[1] = rd<int32_t>(w, 180);
	*count = n;
	if(!objects) { fclose(f); return PT_OK; }          // size query
	if(n > capacity) { fclose(f); return PT_ERR_LIMIT; }
	std::vector<unsigned char> buf(kObjectBytes);
	for(int i = 0; i < n; i++) {
		if(fread(buf.data(), 1, kObjectBytes, f) != kObjectBytes) { fclose(f); return PT_ERR_ARG; }
		const unsigned char* o = buf.data();
		pt_object& d = objects[i];
		memset(&d, 0, sizeof d);
		for(int k = 0; k < 3; k++) {
			d.pos[k] = rd<float>(o, 4 + 4 * k);
			d.dim[k] = rd<float>(o, 32 + 4 * k);
			d.color[k] = rd<float>(o, 44 + 4 * k);
			d.emit_color[k] = rd<float>(o, 76 + 4 * k);
		}
		for(int k = 0; k < 4; k++) d.rot[k] = rd<float>(o, 16 + 4 * k);
		d.type = rd<int32_t>(o, 56);
		d.reflection_mod = rd<float>(o, 88);
		d.refractive_index = rd<float>(o, 92);
		d.roughness = rd<float>(o, 96);
	}
	fclose(f);
	return PT_OK;
}

int pt_scene_save(const char* path, const pt_world* world, const pt_object* objects, int count) {
	if(!path || !world || count < 0 || (count > 0 && !objects)) return PT_ERR_ARG;
	FILE* f = fopen(path, "wb");
	if(!f) return PT_ERR_ARG;
	unsigned char w[kWorldBytes];
	memset(w, 0, sizeof w);
	for(int k = 0; k < 3; k++) {
		wr<float>(w, 0 + 4 * k, world->camera.pos[k]);
		wr<float>(w, 12 + 4 * k, world->camera.rot[k]);
		wr<float>(w, 104 + 4 * k, world->background_color[k]);
		wr<float>(w, 116 + 4 * k, world->ambient_color[k]);
		wr<float>(w, 140 + 4 * k, world->light_color[k]);
	}
	wr<float>(w, 24, world->camera.plane_width);
	wr<float>(w, 36, 10000.0f);                        // farDist, code/raycast.cpp:1302
	wr<float>(w, 40, world->camera.fov);
	wr<int32_t>(w, 88, count);
	wr<int32_t>(w, 92, count);
	wr<int32_t>(w, 96, 100);                           // DArray::startSize, code/rt_misc.cpp:1260
	wr<float>(w, 152, world->focal_distance);
	wr<float>(w, 156, world->aperture);
	wr<int32_t>(w, 176, world->light_rot[0]);
	wr<int32_t>(w, 180, world->light_rot[1]);
	bool ok = fwrite(w, 1, kWorldBytes, f) == kWorldBytes;
	unsigned char o[kObjectBytes];
	for(int i = 0; ok && i < count; i++) {
		const pt_object& s = objects[i];
		memset(o, 0, sizeof o);
		wr<int32_t>(o, 0, i);
		for(int k = 0; k < 3; k++) {
			wr<float>(o, 4 + 4 * k, s.pos[k]);
			wr<float>(o, 32 + 4 * k, s.dim[k]);
			wr<float>(o, 44 + 4 * k, s.color[k]);
			wr<float>(o, 76 + 4 * k, s.emit_color[k]);
		}
		for(int k = 0; k < 4; k++) wr<float>(o, 16 + 4 * k, s.rot[k]);
		wr<int32_t>(o, 56, s.type);
		wr<float>(o, 88, s.reflection_mod);
		wr<float>(o, 92, s.refractive_index);
		wr<float>(o, 96, s.roughness);
		ok = fwrite(o, 1, kObjectBytes, f) == kObjectBytes;
	}
	fclose(f);
	return ok ? PT_OK : PT_ERR_ARG;
}

}  // extern "C"
