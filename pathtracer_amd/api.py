"""ctypes mirror of include/pt_api.h and the loader of libptcore.so (the HIP render core).

Nothing here computes: every call goes through the C ABI.  There is no CPU fallback - if the
library is missing or the device fails, the call raises.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# PT_CORE_LIB selects another build of the same library (A/B timing of kernel variants)
CORE_PATH = os.environ.get("PT_CORE_LIB") or os.path.join(ROOT, "pathtracer_amd", "libptcore.so")
_libs = {}


def _load(path):
    if path not in _libs:
        _libs[path] = C.CDLL(path)
    return _libs[path]

PT_GEOM_SPHERE, PT_GEOM_BOX, PT_GEOM_CYLINDER, PT_GEOM_CONE, PT_GEOM_PLANE = range(5)
PT_SEED_PER_JOB, PT_SEED_PER_SAMPLE = 1, 3
PT_F_STRIDE = 36
PT_W_COUNT = 27
PT_TABLE_COUNT = 20000
F = dict(POS=0, ROT=3, ROTINV=7, DIM=11, SCALE=14, SCALEINV=17, COLOR=20, EMIT=23, BBOX=26,
         BRADIUS=29, REFL=30, IOR=31, ROUGH=32, TYPE=33, FLAGS=34)


class pt_object(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("rot", C.c_float * 4), ("dim", C.c_float * 3),
                ("color", C.c_float * 3), ("type", C.c_int32), ("emit_color", C.c_float * 3),
                ("reflection_mod", C.c_float), ("refractive_index", C.c_float),
                ("roughness", C.c_float)]


class pt_camera(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("rot", C.c_float * 3), ("fov", C.c_float),
                ("plane_width", C.c_float)]


class pt_world(C.Structure):
    _fields_ = [("camera", pt_camera), ("background_color", C.c_float * 3),
                ("ambient_color", C.c_float * 3), ("light_color", C.c_float * 3),
                ("light_rot", C.c_int32 * 2), ("focal_distance", C.c_float),
                ("aperture", C.c_float)]


class pt_settings(C.Structure):
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("spp_wanted", C.c_int32),
                ("sample_grid_width", C.c_int32), ("max_bounces", C.c_int32),
                ("grid_hint", C.c_int32), ("brute_force_below", C.c_int32),
                ("seed_policy", C.c_int32), ("seed", C.c_uint64), ("part_index", C.c_int32),
                ("part_count", C.c_int32), ("frame_index", C.c_int32), ("sample_ranges", C.c_int32)]


class pt_progress(C.Structure):
    _fields_ = [("done", C.c_int32), ("aborted", C.c_int32), ("fraction", C.c_float),
                ("seconds", C.c_double), ("paths_done", C.c_uint64)]


class pt_stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("pixels", C.c_uint64),
                ("zero_sample_pixels", C.c_uint64), ("samples_total", C.c_int32),
                ("objects", C.c_int32), ("cells", C.c_int32 * 3), ("cell_size", C.c_int32 * 3),
                ("cell_items", C.c_int32), ("grid_blocks", C.c_int32), ("trace_ms", C.c_float),
                ("dominant_kernel_ms", C.c_float), ("prepare_ms", C.c_float),
                ("trace_ms_min", C.c_float), ("trace_ms_max", C.c_float), ("trace_launches", C.c_int32)]


def default_settings(width=64, height=36, spp=16, bounces=8, seed=12345, **kw):
    s = pt_settings()
    s.width, s.height, s.spp_wanted = width, height, spp
    s.sample_grid_width, s.max_bounces, s.grid_hint = 3, bounces, 10
    s.brute_force_below, s.seed_policy, s.seed = 10, PT_SEED_PER_SAMPLE, seed
    s.part_index, s.part_count, s.frame_index = 0, 1, -1
    for k, v in kw.items():
        setattr(s, k, v)
    return s


def make_object(pos, dim, type=PT_GEOM_SPHERE, rot=(1, 0, 0, 0), color=(0.5, 0.5, 0.5),
                emit=(0, 0, 0), refl=0.0, ior=1.0, rough=0.0):
    o = pt_object()
    o.pos[:] = pos
    o.rot[:] = rot
    o.dim[:] = dim
    o.color[:] = color
    o.type = type
    o.emit_color[:] = emit
    o.reflection_mod, o.refractive_index, o.roughness = refl, ior, rough
    return o


def make_world(cam_pos=(0, -30, 7), cam_rot=(0, 0, 0), fov=90.0, plane_width=10.0,
               bg=(1, 1, 1), ambient=(1, 1, 1), light=(1, 1, 1), light_rot=(0, 90),
               focal_distance=30.0, aperture=0.0):
    w = pt_world()
    w.camera.pos[:] = cam_pos
    w.camera.rot[:] = cam_rot
    w.camera.fov, w.camera.plane_width = fov, plane_width
    w.background_color[:] = bg
    w.ambient_color[:] = ambient
    w.light_color[:] = light
    w.light_rot[:] = light_rot
    w.focal_distance, w.aperture = focal_distance, aperture
    return w


def object_array(objs):
    arr = (pt_object * max(1, len(objs)))()
    for i, o in enumerate(objs):
        arr[i] = o
    return arr


def fptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def iptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


def uptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint32))


def u64ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint64))


def u8ptr(a):
    return a.ctypes.data_as(C.POINTER(C.c_uint8))


def load_core():
    if not os.path.exists(CORE_PATH):
        raise RuntimeError("libptcore.so is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    lib = _load(CORE_PATH)
    lib.pt_create.restype = C.c_void_p
    lib.pt_create.argtypes = [C.c_int]
    lib.pt_destroy.argtypes = [C.c_void_p]
    lib.pt_last_error.restype = C.c_char_p
    lib.pt_last_error.argtypes = [C.c_void_p]
    lib.pt_render_start.argtypes = [C.c_void_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int,
                                    C.POINTER(pt_settings), C.POINTER(C.c_float)]
    lib.pt_render_progress.argtypes = [C.c_void_p, C.POINTER(pt_progress)]
    lib.pt_render_abort.argtypes = [C.c_void_p]
    lib.pt_render_wait.argtypes = [C.c_void_p]
    lib.pt_prepare.argtypes = [C.c_void_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int,
                               C.POINTER(pt_settings)]
    lib.pt_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
    lib.pt_set_frame.argtypes = [C.c_void_p, C.c_int]
    lib.pt_download.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.pt_get_stats.argtypes = [C.c_void_p, C.POINTER(pt_stats)]
    lib.pt_debug_primary.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint64)]
    lib.pt_cast_rays.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int,
                                 C.POINTER(C.c_int32), C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pt_group_create.restype = C.c_void_p
    lib.pt_group_create.argtypes = [C.POINTER(C.c_int), C.c_int]
    lib.pt_group_destroy.argtypes = [C.c_void_p]
    lib.pt_group_size.argtypes = [C.c_void_p]
    lib.pt_group_ctx.restype = C.c_void_p
    lib.pt_group_ctx.argtypes = [C.c_void_p, C.c_int]
    lib.pt_group_last_error.restype = C.c_char_p
    lib.pt_group_last_error.argtypes = [C.c_void_p]
    lib.pt_group_render_start.argtypes = [C.c_void_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int,
                                          C.POINTER(pt_settings), C.POINTER(C.c_float)]
    lib.pt_group_render_progress.argtypes = [C.c_void_p, C.POINTER(pt_progress)]
    lib.pt_group_render_snapshot.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    lib.pt_group_render_abort.argtypes = [C.c_void_p]
    lib.pt_group_render_wait.argtypes = [C.c_void_p]
    lib.pt_group_prepare.argtypes = [C.c_void_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int, C.POINTER(pt_settings)]
    lib.pt_group_set_frame.argtypes = [C.c_void_p, C.c_int]
    lib.pt_group_trace.argtypes = [C.c_void_p, C.c_int]
    lib.pt_group_sync.argtypes = [C.c_void_p]
    lib.pt_group_download.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.pt_image_alloc.restype = C.POINTER(C.c_float)
    lib.pt_image_alloc.argtypes = [C.c_void_p, C.c_int, C.c_int]
    lib.pt_image_free.argtypes = [C.c_void_p, C.POINTER(C.c_float)]
    lib.pt_pick_rays.argtypes = lib.pt_cast_rays.argtypes
    lib.pt_cast_rays_noalloc.argtypes = lib.pt_cast_rays.argtypes
    lib.pt_debug_tile_cost.argtypes = [C.c_void_p, C.POINTER(C.c_uint16), C.c_int, C.POINTER(C.c_int)]
    lib.pt_cast_histogram.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.c_int]
    lib.pt_debug_device_memory.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pt_render_snapshot.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.POINTER(C.c_uint64)]
    lib.pt_present_srgb8.argtypes = [C.c_void_p, C.POINTER(C.c_uint8)]
    lib.pt_debug_srgb8.argtypes = [C.c_void_p, C.POINTER(C.c_float), C.c_size_t, C.POINTER(C.c_uint8)]
    lib.pt_debug_srgb8_sweep.argtypes = [C.c_void_p, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.pt_host_srgb8_thresholds.argtypes = [C.POINTER(C.c_float)]
    lib.pt_debug_arith.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_float)]
    lib.pt_debug_arith_sweep.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_uint64, C.POINTER(C.c_uint64)]
    lib.pt_default_scene.argtypes = [C.POINTER(pt_world), C.POINTER(pt_object), C.c_int, C.POINTER(C.c_int)]
    lib.pt_default_settings.argtypes = [C.POINTER(pt_settings)]
    lib.pt_host_flatten.argtypes = [C.POINTER(pt_object), C.c_int, C.POINTER(C.c_float)]
    lib.pt_host_world.argtypes = [C.POINTER(pt_world), C.c_int, C.c_int, C.POINTER(C.c_float)]
    lib.pt_host_tables.argtypes = [C.c_uint64, C.c_int, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float),
                                   C.POINTER(C.c_float), C.c_int, C.POINTER(C.c_int32), C.POINTER(C.c_int)]
    lib.pt_host_grid.argtypes = [C.POINTER(C.c_float), C.c_int, C.c_int, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_int,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pt_host_list_entry.argtypes = [C.POINTER(C.c_float), C.c_float, C.c_uint32, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    lib.pt_host_cell_order.argtypes = [C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.pt_host_filter_sphere.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pt_host_cone_slopes.argtypes = [C.POINTER(C.c_float), C.POINTER(C.c_float)]
    lib.pt_sync.argtypes = [C.c_void_p]
    lib.pt_debug_grid.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32), C.c_int, C.POINTER(C.c_uint32),
                                  C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    lib.pt_sample_ranges.argtypes = [C.POINTER(pt_settings)]
    lib.pt_partition_tiles.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int]
    lib.pt_scene_load.argtypes = [C.c_char_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int, C.POINTER(C.c_int)]
    lib.pt_scene_save.argtypes = [C.c_char_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int]
    lib.pt_partition_floats.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
    lib.pt_assemble.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    return lib


class PtError(RuntimeError):
    pass


class Renderer:
    """Thin host-side mirror of the C ABI: one context on one GPU.  Every method is a direct call
    into libptcore.so; errors raise PtError with the library's message."""

    def __init__(self, device=0):
        self.lib = load_core()
        self.ctx = self.lib.pt_create(device)
        if not self.ctx:
            raise PtError("pt_create failed: " + self.lib.pt_last_error(None).decode())
        self.settings = None

    def close(self):
        if self.ctx:
            self.lib.pt_destroy(self.ctx)
            self.ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc, what):
        if rc != 0:
            raise PtError(f"{what} failed ({rc}): {self.lib.pt_last_error(self.ctx).decode()}")

    # --- render start / progress / abort / wait (the reference's boundary) ---
    def start(self, world, objs, settings, out=None):
        self.settings = settings
        self._objs = object_array(objs)
        self._out = out if out is not None else np.zeros((settings.height, settings.width, 3), np.float32)
        self._check(self.lib.pt_render_start(self.ctx, C.byref(world), self._objs, len(objs), C.byref(settings),
                                             fptr(self._out)), "pt_render_start")
        return self._out

    def image_alloc(self, width, height):
        """A page-locked (height, width, 3) float image owned by the library; release it with image_free."""
        p = self.lib.pt_image_alloc(self.ctx, width, height)
        if not p:
            raise PtError("pt_image_alloc failed: " + self.lib.pt_last_error(self.ctx).decode())
        a = np.ctypeslib.as_array(p, shape=(height, width, 3))
        self._images = getattr(self, "_images", {})
        self._images[a.ctypes.data] = p
        return a

    def image_free(self, a):
        self.lib.pt_image_free(self.ctx, self._images.pop(a.ctypes.data))

    def progress(self):
        p = pt_progress()
        rc = self.lib.pt_render_progress(self.ctx, C.byref(p))
        self._check(rc, "pt_render_progress")
        return p

    def snapshot(self, out=None):
        """Pixels finished so far copied into `out` (default: the image given to start); returns their count."""
        n = C.c_uint64(0)
        target = self._out if out is None else out
        self._check(self.lib.pt_render_snapshot(self.ctx, fptr(target), C.byref(n)), "pt_render_snapshot")
        return n.value

    def abort(self):
        self._check(self.lib.pt_render_abort(self.ctx), "pt_render_abort")

    def wait(self):
        self._check(self.lib.pt_render_wait(self.ctx), "pt_render_wait")
        return self._out

    def render(self, world, objs, settings):
        self.start(world, objs, settings)
        return self.wait()

    # --- resident-scene interface ---
    def prepare(self, world, objs, settings):
        self.settings = settings
        self._objs = object_array(objs)
        self._check(self.lib.pt_prepare(self.ctx, C.byref(world), self._objs, len(objs), C.byref(settings)), "pt_prepare")

    def trace(self, d_rgb=None, stream=None, sync=True):
        self._check(self.lib.pt_trace(self.ctx, d_rgb, stream, 1 if sync else 0), "pt_trace")

    def sync(self):
        self._check(self.lib.pt_sync(self.ctx), "pt_sync")

    def debug_grid(self):
        """(cell_start, cell_items, obj_range, built_on_device) of the prepared scene's grid."""
        cells, items, dev = C.c_int(0), C.c_int(0), C.c_int(0)
        self._check(self.lib.pt_debug_grid(self.ctx, None, 0, None, 0, None, C.byref(cells), C.byref(items), C.byref(dev)), "pt_debug_grid")
        start = np.zeros(cells.value + 1, np.uint32)
        lists = np.zeros(max(1, items.value), np.uint32)
        ranges = np.zeros(max(1, 2 * self.stats().objects), np.uint32)
        self._check(self.lib.pt_debug_grid(self.ctx, uptr(start), start.size, uptr(lists), lists.size, uptr(ranges),
                                           C.byref(cells), C.byref(items), C.byref(dev)), "pt_debug_grid")
        return start, lists[:items.value], ranges[:2 * self.stats().objects], bool(dev.value)

    def set_frame(self, frame):
        self._check(self.lib.pt_set_frame(self.ctx, frame), "pt_set_frame")

    def download(self):
        s = self.settings
        out = np.zeros((s.height, s.width, 3), np.float32)
        self._check(self.lib.pt_download(self.ctx, fptr(out)), "pt_download")
        return out

    def stats(self):
        st = pt_stats()
        self._check(self.lib.pt_get_stats(self.ctx, C.byref(st)), "pt_get_stats")
        return st

    def partition_floats(self):
        n = C.c_size_t(0)
        self._check(self.lib.pt_partition_floats(self.ctx, C.byref(n)), "pt_partition_floats")
        return n.value

    def assemble(self, d_parts, stride_floats, part_count, width, height, d_rgb, stream=None):
        self._check(self.lib.pt_assemble(self.ctx, d_parts, stride_floats, part_count, width, height, d_rgb, stream), "pt_assemble")

    def debug_primary(self):
        s = self.settings
        ids = np.zeros((s.height, s.width), np.int32)
        hs = np.zeros((s.height, s.width), np.uint64)
        self._check(self.lib.pt_debug_primary(self.ctx, iptr(ids), u64ptr(hs)), "pt_debug_primary")
        return ids, hs

    def cast_rays(self, origins, dirs, pick=False):
        """pick=False: the tracer's cast (pt_cast_rays); True: the editor's picking ray (pt_pick_rays)."""
        o = np.ascontiguousarray(origins, np.float32)
        d = np.ascontiguousarray(dirs, np.float32)
        n = o.shape[0]
        ids = np.zeros(n, np.int32)
        pos = np.zeros((n, 3), np.float32)
        nrm = np.zeros((n, 3), np.float32)
        fn = self.lib.pt_pick_rays if pick else self.lib.pt_cast_rays
        self._check(fn(self.ctx, fptr(o), fptr(d), n, iptr(ids), fptr(pos), fptr(nrm)), "pt_pick_rays" if pick else "pt_cast_rays")
        return ids, pos, nrm

    def arith(self, op, a, b=None):
        """Device arithmetic probe: op 0 sqrt(a), 1 a[i, :] / b[i], 2 a[i, :] / |a[i, :]|, 3 a[i, :] / pi."""
        a = np.ascontiguousarray(a, np.float32)
        n = a.shape[0]
        out = np.zeros_like(a)
        bb = fptr(np.ascontiguousarray(b, np.float32)) if b is not None else None
        self._check(self.lib.pt_debug_arith(self.ctx, op, fptr(a), bb, n, fptr(out)), "pt_debug_arith")
        return out

    def arith_sweep(self, op, first_bits, count):
        bad = C.c_uint64(0)
        self._check(self.lib.pt_debug_arith_sweep(self.ctx, op, first_bits, count, C.byref(bad)), "pt_debug_arith_sweep")
        return bad.value

    def tile_cost(self):
        """Cost estimate of every owned 8x8 tile (pt_debug_tile_cost), in the partition's tile order."""
        n = C.c_int(0)
        self._check(self.lib.pt_debug_tile_cost(self.ctx, None, 0, C.byref(n)), "pt_debug_tile_cost")
        cost = np.zeros(max(1, n.value), np.uint16)
        self._check(self.lib.pt_debug_tile_cost(self.ctx, cost.ctypes.data_as(C.POINTER(C.c_uint16)), cost.size, C.byref(n)), "pt_debug_tile_cost")
        return cost[:n.value]

    def cast_histogram(self, bins=65):
        """hist[k] = camera samples whose path casts k rays (pt_cast_histogram)."""
        hist = np.zeros(bins, np.uint64)
        self._check(self.lib.pt_cast_histogram(self.ctx, u64ptr(hist), bins), "pt_cast_histogram")
        return hist

    def present_srgb8(self):
        s = self.settings
        out = np.zeros((s.height, s.width, 3), np.uint8)
        self._check(self.lib.pt_present_srgb8(self.ctx, u8ptr(out)), "pt_present_srgb8")
        return out


    def srgb8(self, linear):
        """The screenshot encoder on arbitrary float values."""
        a = np.ascontiguousarray(linear, np.float32).ravel()
        out = np.zeros(a.size, np.uint8)
        self._check(self.lib.pt_debug_srgb8(self.ctx, fptr(a), a.size, u8ptr(out)), "pt_debug_srgb8")
        return out

    def srgb8_sweep(self, first_bits, count):
        bad = C.c_uint64(0)
        hist = np.zeros(256, np.uint64)
        self._check(self.lib.pt_debug_srgb8_sweep(self.ctx, first_bits, count, C.byref(bad), u64ptr(hist)), "pt_debug_srgb8_sweep")
        return bad.value, hist


class Group:
    """Mirror of the pt_group_* calls: one frame on several GPUs (or several partitions of one GPU) of this process."""

    def __init__(self, devices):
        self.lib = load_core()
        arr = (C.c_int * len(devices))(*devices)
        self.g = self.lib.pt_group_create(arr, len(devices))
        if not self.g:
            raise PtError("pt_group_create failed: " + self.lib.pt_last_error(None).decode())

    def close(self):
        if self.g:
            self.lib.pt_group_destroy(self.g)
            self.g = None

    def _check(self, rc, what):
        if rc != 0:
            raise PtError(f"{what} failed ({rc}): {self.lib.pt_group_last_error(self.g).decode()}")

    def start(self, world, objs, settings, out=None):
        self._objs = object_array(objs)
        self._out = out if out is not None else np.zeros((settings.height, settings.width, 3), np.float32)
        self._check(self.lib.pt_group_render_start(self.g, C.byref(world), self._objs, len(objs), C.byref(settings), fptr(self._out)),
                    "pt_group_render_start")
        return self._out

    def progress(self):
        p = pt_progress()
        self._check(self.lib.pt_group_render_progress(self.g, C.byref(p)), "pt_group_render_progress")
        return p

    def snapshot(self, out=None):
        n = C.c_uint64(0)
        self._check(self.lib.pt_group_render_snapshot(self.g, fptr(self._out if out is None else out), C.byref(n)), "pt_group_render_snapshot")
        return n.value

    def abort(self):
        self._check(self.lib.pt_group_render_abort(self.g), "pt_group_render_abort")

    def wait(self):
        self._check(self.lib.pt_group_render_wait(self.g), "pt_group_render_wait")
        return self._out

    def render(self, world, objs, settings, out=None):
        self.start(world, objs, settings, out)
        return self.wait()

    # --- resident frames ---
    def prepare(self, world, objs, settings):
        self._objs = object_array(objs)
        self._settings = settings
        self._check(self.lib.pt_group_prepare(self.g, C.byref(world), self._objs, len(objs), C.byref(settings)), "pt_group_prepare")

    def set_frame(self, frame):
        self._check(self.lib.pt_group_set_frame(self.g, frame), "pt_group_set_frame")

    def trace(self, sync=True):
        self._check(self.lib.pt_group_trace(self.g, 1 if sync else 0), "pt_group_trace")

    def sync(self):
        self._check(self.lib.pt_group_sync(self.g), "pt_group_sync")

    def download(self):
        s = self._settings
        out = np.zeros((s.height, s.width, 3), np.float32)
        self._check(self.lib.pt_group_download(self.g, fptr(out)), "pt_group_download")
        return out

    def part_stats(self, part):
        st = pt_stats()
        rc = self.lib.pt_get_stats(self.lib.pt_group_ctx(self.g, part), C.byref(st))
        if rc != 0:
            raise PtError(f"pt_get_stats failed ({rc})")
        return st


def srgb8_thresholds():
    out = np.zeros(255, np.float32)
    rc = load_core().pt_host_srgb8_thresholds(fptr(out))
    if rc != 0:
        raise PtError(f"pt_host_srgb8_thresholds failed ({rc})")
    return out


def save_scene(path, world, objs):
    """Writes a scene file in the reference application's format."""
    rc = load_core().pt_scene_save(path.encode(), C.byref(world), object_array(objs), len(objs))
    if rc != 0:
        raise PtError(f"pt_scene_save({path}) failed ({rc})")


def load_scene(path):
    lib = load_core()
    world, n = pt_world(), C.c_int(0)
    rc = lib.pt_scene_load(path.encode(), C.byref(world), None, 0, C.byref(n))
    if rc != 0:
        raise PtError(f"pt_scene_load({path}) failed ({rc})")
    arr = (pt_object * max(1, n.value))()
    rc = lib.pt_scene_load(path.encode(), C.byref(world), arr, n.value, C.byref(n))
    if rc != 0:
        raise PtError(f"pt_scene_load({path}) failed ({rc})")
    return world, [arr[i] for i in range(n.value)]
