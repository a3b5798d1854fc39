// Headless Linux harness for the render core: plays the part of the reference's frame function
// (code/app.cpp:1127-1461) without a window.  It starts a render, polls progress once per "frame",
// optionally aborts, waits, and writes the image - all through the C ABI of include/pt_api.h, the
// same calls the OpenGL editor would make in place of its start / poll / abort blocks.
//
//   pt_headless [--scene default|FILE.scene] [--size WxH] [--spp N] [--bounces N] [--grid-hint N]
//               [--seed N] [--gpus 0,1,...] [--out IMAGE.ppm] [--pfm IMAGE.pfm] [--poll-ms N]
//               [--abort-after SECONDS] [--progressive FRAMES] [--bounce-histogram] [--quiet]
//
// --progressive FRAMES plays the interactive mode instead of one render: the scene stays resident (pt_group_prepare), every
// frame traces one more sample per pixel into the accumulation buffers (pt_group_set_frame, pt_group_trace) and the loop
// waits for it like a display would; the JSON line then carries the median and the 99th percentile of the frame times.
// --bounce-histogram adds "casts_per_path": how many camera samples' paths cast 1, 2, ... rays (pt_cast_histogram).
// With several GPUs (--gpus 0,1,...; an ordinal may repeat) the frame goes through the pt_group_* calls: every
// entry gets a context and a round-robin share of the image's 8x8 tiles, the shares are gathered device to device
// on the first GPU, assembled there and copied out once.  Prints one JSON line with the metrics the
// reference shows in its stats panel (code/app.cpp:2879-2896): pixels, samples, time, plus Mpaths/s.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "pt_api.h"

static void usage() {
	fprintf(stderr, "usage: pt_headless [--scene default|FILE.scene] [--size WxH] [--spp N] [--bounces N] [--grid-hint N]\n"
	                "                   [--seed N] [--gpus 0,1,...] [--out IMAGE.ppm] [--pfm IMAGE.pfm] [--poll-ms N]\n"
	                "                   [--abort-after SECONDS] [--progressive FRAMES] [--bounce-histogram] [--quiet]\n");
}

// Screenshot encoding of the reference (code/app.cpp:1541-1547, code/rt_math.cpp:3634-3640).
static unsigned char encode_srgb8(float v) {
	if(v < 0.0031308f) v *= 12.92f;
	else v = 1.055f * powf(v, 1.0f / 2.4f) - 0.055f;
	return (unsigned char)(char)(int)roundf(v * 255);
}

int main(int argc, char** argv) {
	std::string scene = "default", out_ppm, out_pfm;
	std::vector<int> gpus = { 0 };
	pt_settings st;
	pt_default_settings(&st);
	int poll_ms = 16;
	double abort_after = -1;
	bool quiet = false, histogram = false;
	int progressive = 0;
	for(int i = 1; i < argc; i++) {
		std::string a = argv[i];
		auto next = [&]() -> const char* { if(i + 1 >= argc) { usage(); exit(2); } return argv[++i]; };
		if(a == "--scene") scene = next();
		else if(a == "--size") { if(sscanf(next(), "%dx%d", &st.width, &st.height) != 2) { usage(); return 2; } }
		else if(a == "--spp") st.spp_wanted = atoi(next());
		else if(a == "--bounces") st.max_bounces = atoi(next());
		else if(a == "--grid-hint") st.grid_hint = atoi(next());
		else if(a == "--seed") st.seed = strtoull(next(), 0, 10);
		else if(a == "--out") out_ppm = next();
		else if(a == "--pfm") out_pfm = next();
		else if(a == "--poll-ms") poll_ms = atoi(next());
		else if(a == "--abort-after") abort_after = atof(next());
		else if(a == "--quiet") quiet = true;
		else if(a == "--progressive") progressive = atoi(next());
		else if(a == "--bounce-histogram") histogram = true;
		else if(a == "--gpus") {
			gpus.clear();
			std::string l = next();
			for(size_t p = 0; p < l.size();) { size_t q = l.find(',', p); if(q == std::string::npos) q = l.size(); gpus.push_back(atoi(l.substr(p, q - p).c_str())); p = q + 1; }
		}
		else { usage(); return 2; }
	}

	pt_world world;
	std::vector<pt_object> objects;
	if(scene == "default") {
		objects.resize(2);
		int n = 0;
		if(pt_default_scene(&world, objects.data(), 2, &n) != PT_OK) { fprintf(stderr, "pt_default_scene failed\n"); return 1; }
	} else {
		int n = 0;
		if(pt_scene_load(scene.c_str(), &world, nullptr, 0, &n) != PT_OK) { fprintf(stderr, "cannot read scene file %s\n", scene.c_str()); return 1; }
		objects.resize((size_t)(n > 0 ? n : 1));
		if(pt_scene_load(scene.c_str(), &world, objects.data(), n, &n) != PT_OK) { fprintf(stderr, "cannot read scene file %s\n", scene.c_str()); return 1; }
		objects.resize((size_t)n);
	}

	const int parts = (int)gpus.size();
	pt_group* group = pt_group_create(gpus.data(), parts);
	if(!group) { fprintf(stderr, "pt_group_create: %s\n", pt_last_error(nullptr)); return 1; }
	std::vector<float> image((size_t)st.width * st.height * 3, 0.0f);

	// the bounce histogram of the frame: a pass of its own over every partition's paths, after the render
	auto histogram_json = [&]() -> std::string {
		std::vector<unsigned long long> total(65, 0);
		for(int p = 0; p < parts; p++) {
			uint64_t h[65];
			pt_ctx* c = pt_group_ctx(group, p);
			if(pt_cast_histogram(c, h, 65) != PT_OK) { fprintf(stderr, "pt_cast_histogram: %s\n", pt_last_error(c)); exit(1); }
			for(int k = 0; k < 65; k++) total[(size_t)k] += h[k];
		}
		int last = 64;
		while(last > 1 && total[(size_t)last] == 0) last--;
		std::string text = ", \"casts_per_path\": [";
		for(int k = 1; k <= last; k++) text += (k > 1 ? ", " : "") + std::to_string(total[(size_t)k]);
		return text + "]";
	};

	if(progressive > 0) {
		// the interactive mode: one sample per pixel and frame, the loop waits for every frame as a display would
		st.frame_index = 0;
		if(pt_group_prepare(group, &world, objects.data(), (int)objects.size(), &st) != PT_OK) { fprintf(stderr, "pt_group_prepare: %s\n", pt_group_last_error(group)); return 1; }
		std::vector<double> ms;
		auto t_all = std::chrono::steady_clock::now();
		for(int f = 0; f < progressive; f++) {
			auto t_frame = std::chrono::steady_clock::now();
			if(pt_group_set_frame(group, f) != PT_OK || pt_group_trace(group, 1) != PT_OK) { fprintf(stderr, "frame %d: %s\n", f, pt_group_last_error(group)); return 1; }
			ms.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_frame).count());
		}
		const double seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_all).count();
		if(pt_group_download(group, image.data()) != PT_OK) { fprintf(stderr, "pt_group_download: %s\n", pt_group_last_error(group)); return 1; }
		unsigned long long paths = 0, pixels = 0;
		for(int p = 0; p < parts; p++) { pt_stats stats; pt_get_stats(pt_group_ctx(group, p), &stats); paths += stats.paths; pixels += stats.pixels; }
		std::vector<double> sorted = ms;
		std::sort(sorted.begin(), sorted.end());
		const double median = sorted[sorted.size() / 2], p99 = sorted[std::min(sorted.size() - 1, (size_t)(sorted.size() * 0.99))];
		if(!out_pfm.empty()) {
			FILE* f = fopen(out_pfm.c_str(), "wb");
			if(!f) { fprintf(stderr, "cannot write %s\n", out_pfm.c_str()); return 1; }
			fprintf(f, "PF\n%d %d\n-1.0\n", st.width, st.height);
			for(int y = st.height - 1; y >= 0; y--) fwrite(image.data() + (size_t)y * st.width * 3, sizeof(float), (size_t)st.width * 3, f);
			fclose(f);
		}
		printf("{\"scene\": \"%s\", \"mode\": \"progressive\", \"width\": %d, \"height\": %d, \"spp_wanted\": %d, \"max_bounces\": %d, \"objects\": %d, \"gpus\": %d, "
		       "\"pixels\": %llu, \"frames\": %d, \"paths_per_frame\": %llu, \"seconds\": %.6f, \"frame_ms_median\": %.4f, \"frame_ms_p99\": %.4f, \"frame_ms_first\": %.4f%s}\n",
		       scene.c_str(), st.width, st.height, st.spp_wanted, st.max_bounces, (int)objects.size(), parts, pixels, progressive, paths, seconds,
		       median, p99, ms[0], histogram ? histogram_json().c_str() : "");
		pt_group_destroy(group);
		return 0;
	}

	auto t0 = std::chrono::steady_clock::now();
	if(pt_group_render_start(group, &world, objects.data(), (int)objects.size(), &st, image.data()) != PT_OK) {
		fprintf(stderr, "pt_group_render_start: %s\n", pt_group_last_error(group));
		return 1;
	}
	// the application's frame loop: poll once per frame, abort on request
	bool aborted = false;
	for(;;) {
		pt_progress info;
		if(pt_group_render_progress(group, &info) != PT_OK) { fprintf(stderr, "render failed: %s\n", pt_group_last_error(group)); return 1; }
		double elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		if(!quiet) fprintf(stderr, "\r%5.1f %%  %.3f s", info.fraction * 100, elapsed);
		if(info.done) break;
		if(abort_after >= 0 && !aborted && elapsed >= abort_after) {
			pt_group_render_abort(group);
			aborted = true;
		}
		std::this_thread::sleep_for(std::chrono::milliseconds(poll_ms));
	}
	if(!quiet) fprintf(stderr, "\n");
	if(pt_group_render_wait(group) != PT_OK) { fprintf(stderr, "render failed: %s\n", pt_group_last_error(group)); return 1; }
	pt_progress info;
	pt_group_render_progress(group, &info);
	unsigned long long paths = 0, rays = 0, pixels = 0;
	double seconds = info.seconds, kernel_ms = 0;
	int was_aborted = info.aborted;
	for(int p = 0; p < parts; p++) {
		pt_stats stats;
		pt_get_stats(pt_group_ctx(group, p), &stats);
		paths += stats.paths; rays += stats.rays; pixels += stats.pixels;
		kernel_ms = stats.trace_ms > kernel_ms ? stats.trace_ms : kernel_ms;
	}
	if(was_aborted) paths = info.paths_done;

	if(!out_ppm.empty()) {
		FILE* f = fopen(out_ppm.c_str(), "wb");
		if(!f) { fprintf(stderr, "cannot write %s\n", out_ppm.c_str()); return 1; }
		fprintf(f, "P6\n%d %d\n255\n", st.width, st.height);
		std::vector<unsigned char> bytes(image.size());
		if(parts == 1 && !was_aborted) {
			pt_ctx* only = pt_group_ctx(group, 0);
			if(pt_present_srgb8(only, bytes.data()) != PT_OK) { fprintf(stderr, "pt_present_srgb8: %s\n", pt_last_error(only)); return 1; }
		} else {
			for(size_t i = 0; i < image.size(); i++) bytes[i] = encode_srgb8(image[i]);
		}
		fwrite(bytes.data(), 1, bytes.size(), f);
		fclose(f);
	}
	if(!out_pfm.empty()) {
		FILE* f = fopen(out_pfm.c_str(), "wb");
		if(!f) { fprintf(stderr, "cannot write %s\n", out_pfm.c_str()); return 1; }
		fprintf(f, "PF\n%d %d\n-1.0\n", st.width, st.height);
		for(int y = st.height - 1; y >= 0; y--) fwrite(image.data() + (size_t)y * st.width * 3, sizeof(float), (size_t)st.width * 3, f);
		fclose(f);
	}
	printf("{\"scene\": \"%s\", \"width\": %d, \"height\": %d, \"spp_wanted\": %d, \"max_bounces\": %d, \"objects\": %d, \"gpus\": %d, "
	       "\"pixels\": %llu, \"paths\": %llu, \"rays\": %llu, \"seconds\": %.6f, \"kernel_ms\": %.3f, \"mpaths_per_s\": %.1f, \"aborted\": %d%s}\n",
	       scene.c_str(), st.width, st.height, st.spp_wanted, st.max_bounces, (int)objects.size(), parts, pixels, paths, rays, seconds,
	       kernel_ms, seconds > 0 ? paths / seconds / 1e6 : 0.0, was_aborted, histogram && !was_aborted ? histogram_json().c_str() : "");
	pt_group_destroy(group);
	return 0;
}
