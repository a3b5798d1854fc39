"""TEST INFRASTRUCTURE: ctypes loaders for the CPU oracle (oracle/libptoracle.so) and, where it was
built, the reference's own code (oracle/_ref/libptref.so).  Only tests/, __graft_entry__.smoke() and
the cpu_baseline leg of bench.py import this module."""
import ctypes as C
import os

import numpy as np

from pathtracer_amd.api import pt_object, pt_settings, pt_world

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_PATH = os.path.join(ROOT, "oracle", "_ref", "libptref.so")
ORACLE_PATH = os.path.join(ROOT, "oracle", "libptoracle.so")
_libs = {}

FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int32)
UP = C.POINTER(C.c_uint32)
U64P = C.POINTER(C.c_uint64)
U8P = C.POINTER(C.c_uint8)
VP = C.c_void_p


def _load(path):
    if path not in _libs:
        _libs[path] = C.CDLL(path)
    return _libs[path]


def have_ref():
    return os.path.exists(REF_PATH)


def _sig(lib, name, restype, *argtypes):
    fn = getattr(lib, name)
    fn.restype = restype
    fn.argtypes = list(argtypes)


def load_ref():
    lib = _load(REF_PATH)
    _sig(lib, "ref_pcg_seed", None, C.c_uint64, C.c_uint64)
    _sig(lib, "ref_pcg_u32", None, C.c_int, UP)
    _sig(lib, "ref_pcg_int", None, C.c_int, C.c_int, C.c_int, IP)
    _sig(lib, "ref_pcg_float", None, C.c_float, C.c_float, C.c_float, C.c_int, FP)
    _sig(lib, "ref_tables", None, FP, FP)
    _sig(lib, "ref_scene_setup", C.c_int, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int,
         C.POINTER(pt_settings), C.c_int)
    _sig(lib, "ref_get_samples", None, FP, IP)
    _sig(lib, "ref_get_records", None, FP)
    _sig(lib, "ref_get_world", None, FP)
    _sig(lib, "ref_get_grid", None, IP)
    _sig(lib, "ref_get_cell", C.c_int, C.c_int, IP, C.c_int)
    _sig(lib, "ref_line_sphere", None, C.c_int, FP, FP, FP, FP, FP, FP, FP)
    _sig(lib, "ref_line_box", None, C.c_int, FP, FP, FP, FP, FP, FP, FP)
    _sig(lib, "ref_line_cylinder", None, C.c_int, FP, FP, FP, FP, FP)
    _sig(lib, "ref_line_cone", None, C.c_int, FP, FP, FP, FP, FP)
    _sig(lib, "ref_sphere_collision", None, C.c_int, FP, FP, FP, FP, IP)
    _sig(lib, "ref_optics", None, C.c_int, FP, FP, FP, FP, FP, FP)
    _sig(lib, "ref_get_intersection", None, C.c_int, IP, FP, FP, FP, FP, FP)
    _sig(lib, "ref_traverse", C.c_int, FP, FP, C.c_int, IP, IP)
    _sig(lib, "ref_cast", None, C.c_int, FP, FP, C.c_int, IP, FP, FP)
    _sig(lib, "ref_jobs", C.c_int, IP)
    _sig(lib, "ref_render", C.c_int, C.c_uint64, FP)
    _sig(lib, "ref_render_mt", C.c_int, C.c_uint64, C.c_int, FP)
    _sig(lib, "ref_render_rect", None, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, FP)
    _sig(lib, "ref_primary_hits", None, IP)
    _sig(lib, "ref_srgb8", None, C.c_int, FP, U8P)
    _sig(lib, "ref_sizeof", C.c_int, C.c_int)
    _sig(lib, "ref_save_scene", None, C.c_char_p)
    _sig(lib, "ref_load_scene", C.c_int, C.c_char_p, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int)
    return lib


def load_oracle():
    lib = _load(ORACLE_PATH)
    _sig(lib, "pto_pcg_seed", None, C.c_uint64, C.c_uint64)
    _sig(lib, "pto_pcg_u32", None, C.c_int, UP)
    _sig(lib, "pto_pcg_int", None, C.c_int, C.c_int, C.c_int, IP)
    _sig(lib, "pto_pcg_float", None, C.c_float, C.c_float, C.c_float, C.c_int, FP)
    _sig(lib, "pto_scene_create", VP, C.POINTER(pt_world), C.POINTER(pt_object), C.c_int,
         C.POINTER(pt_settings))
    _sig(lib, "pto_scene_destroy", None, VP)
    _sig(lib, "pto_get_tables", None, VP, FP, FP)
    _sig(lib, "pto_get_samples", C.c_int, VP, FP, IP)
    _sig(lib, "pto_get_records", None, VP, FP)
    _sig(lib, "pto_get_world", None, VP, FP)
    _sig(lib, "pto_get_grid", None, VP, IP)
    _sig(lib, "pto_get_cell", C.c_int, VP, C.c_int, IP, C.c_int)
    _sig(lib, "pto_get_csr", C.c_int, VP, UP, UP)
    _sig(lib, "pto_line_sphere", None, C.c_int, FP, FP, FP, FP, FP, FP, FP)
    _sig(lib, "pto_line_box", None, C.c_int, FP, FP, FP, FP, FP, FP, FP)
    _sig(lib, "pto_line_cylinder", None, C.c_int, FP, FP, FP, FP, FP)
    _sig(lib, "pto_line_cone", None, C.c_int, FP, FP, FP, FP, FP)
    _sig(lib, "pto_sphere_collision", None, C.c_int, FP, FP, FP, FP, IP)
    _sig(lib, "pto_optics", None, C.c_int, FP, FP, FP, FP, FP, FP)
    _sig(lib, "pto_get_intersection", None, VP, C.c_int, IP, FP, FP, FP, FP, FP)
    _sig(lib, "pto_traverse", C.c_int, VP, FP, FP, C.c_int, IP, IP)
    _sig(lib, "pto_cast", None, VP, C.c_int, FP, FP, C.c_int, IP, FP, FP)
    _sig(lib, "pto_jobs", C.c_int, VP, IP, C.c_int)
    _sig(lib, "pto_render", C.c_int, VP, C.c_uint64, C.c_int, FP)
    _sig(lib, "pto_render_rect", None, VP, C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, FP)
    _sig(lib, "pto_render_mt", C.c_int, VP, C.c_uint64, C.c_int, C.c_int, FP)
    _sig(lib, "pto_primary", None, VP, IP, U64P)
    _sig(lib, "pto_sample_radiance", C.c_int, VP, C.c_int, C.c_int, FP, C.c_int)
    _sig(lib, "pto_progressive_frame", None, VP, C.c_int, FP)
    _sig(lib, "pto_path_count", C.c_uint64, VP)
    _sig(lib, "pto_counters", None, VP, U64P)
    _sig(lib, "pto_counters_reset", None, VP)
    _sig(lib, "pto_cast_histogram", None, VP, U64P, C.c_int)
    _sig(lib, "pto_srgb8", None, C.c_int, FP, U8P)
    _sig(lib, "pto_srgb8_check", C.c_uint64, FP, C.c_uint32, C.c_uint64)
    return lib


COUNTER_NAMES = ("casts", "dda_steps", "bsphere_tests", "isect_sphere", "isect_box", "isect_cylinder",
                 "isect_cone", "isect_plane", "rotated_box", "uneven_sphere", "shade_opaque",
                 "shade_oren", "shade_glass", "paths", "lens", "emit")


class OracleScene:
    """Owns one pto_scene."""

    def __init__(self, world, objects, settings):
        from pathtracer_amd.api import object_array
        self.lib = load_oracle()
        self.world, self.settings = world, settings
        self.n = len(objects)
        self._objs = object_array(objects)
        self.h = self.lib.pto_scene_create(C.byref(world), self._objs, self.n, C.byref(settings))
        self.W, self.H = settings.width, settings.height

    def close(self):
        if self.h:
            self.lib.pto_scene_destroy(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def render(self, policy=3, seed_base=0, threads=0):
        rgb = np.zeros((self.H, self.W, 3), np.float32)
        p = rgb.ctypes.data_as(FP)
        if threads > 0:
            self.lib.pto_render_mt(self.h, seed_base, policy, threads, p)
        else:
            self.lib.pto_render(self.h, seed_base, policy, p)
        return rgb

    def primary(self):
        ids = np.zeros((self.H, self.W), np.int32)
        hs = np.zeros((self.H, self.W), np.uint64)
        self.lib.pto_primary(self.h, ids.ctypes.data_as(IP), hs.ctypes.data_as(U64P))
        return ids, hs

    def samples(self):
        gw = self.settings.sample_grid_width
        offs = np.zeros(gw * gw + 1, np.int32)
        n = self.lib.pto_get_samples(self.h, None, offs.ctypes.data_as(IP))
        smp = np.zeros((max(n, 1), 2), np.float32)
        self.lib.pto_get_samples(self.h, smp.ctypes.data_as(FP), offs.ctypes.data_as(IP))
        return smp[:n], offs

    def records(self):
        r = np.zeros((max(self.n, 1), 36), np.float32)
        self.lib.pto_get_records(self.h, r.ctypes.data_as(FP))
        return r[:self.n]

    def world_consts(self):
        o = np.zeros(27, np.float32)
        self.lib.pto_get_world(self.h, o.ctypes.data_as(FP))
        return o

    def grid(self):
        g = np.zeros(9, np.int32)
        self.lib.pto_get_grid(self.h, g.ctypes.data_as(IP))
        cells = int(g[3]) * int(g[4]) * int(g[5])
        start = np.zeros(cells + 1, np.uint32)
        n = self.lib.pto_get_csr(self.h, start.ctypes.data_as(UP), None)
        items = np.zeros(max(n, 1), np.uint32)
        self.lib.pto_get_csr(self.h, start.ctypes.data_as(UP), items.ctypes.data_as(UP))
        return g, start, items[:n]

    def tables(self):
        d = np.zeros((20000, 3), np.float32)
        c = np.zeros((20000, 2), np.float32)
        self.lib.pto_get_tables(self.h, d.ctypes.data_as(FP), c.ctypes.data_as(FP))
        return d, c

    def path_count(self):
        return int(self.lib.pto_path_count(self.h))

    def counters(self):
        c = np.zeros(16, np.uint64)
        self.lib.pto_counters(self.h, c.ctypes.data_as(U64P))
        return dict(zip(COUNTER_NAMES, (int(v) for v in c)))

    def cast_histogram(self, bins=65):
        h = np.zeros(bins, np.uint64)
        self.lib.pto_cast_histogram(self.h, h.ctypes.data_as(U64P), bins)
        return h

    def sample_radiance(self, x, y, cap=8192):
        out = np.zeros((cap, 3), np.float32)
        n = self.lib.pto_sample_radiance(self.h, x, y, out.ctypes.data_as(FP), cap)
        return out[:n]
