"""ctypes binding of the C ABI (include/xpbd.h) and of the host mirror's C window.

Test / bench glue only: the product is the shared library.  There is no Python or
CPU fallback -- if libxpbd_hip.so is missing the import fails, and without a GPU
every compute call returns an error that is raised as XpbdError.
"""
import ctypes as C
import os

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_DIR = os.path.join(_PKG, "lib")

RIGID_DOUBLES = 38
BYTES_PER_BODY_SUBSTEP = 412  # read 13+25 doubles + 4 B shape id, write 13 doubles

MODE_FUSED = 0
MODE_PER_SUBSTEP = 1
MODE_CONTACTS = 2  # extension: ground + body-body contacts
FLAG_TRACE_CONTACTS = 1

OK, E_INVALID, E_HIP, E_OOM, E_SINGULAR_INERTIA, E_NO_DEVICE, E_CAPACITY, E_HALO = 0, -1, -2, -3, -4, -5, -6, -7

# xpbd_rigid field -> (first double, count); order of reference src/rigid.rs:6-50
RIGID_FIELDS = {
    "inverse_mass": (0, 1), "inverse_inertia": (1, 9), "external_force": (10, 3), "internal_force": (13, 3),
    "external_torque": (16, 3), "internal_torque": (19, 3), "velocity": (22, 3), "angular_velocity": (25, 3),
    "center_of_mass": (28, 3), "position": (31, 3), "rotation": (34, 4),
}

# Every symbol include/xpbd.h declares (tests check the library exports all of them).
ABI_SYMBOLS = [
    "xpbd_abi_version", "xpbd_last_error", "xpbd_config_default", "xpbd_device_count", "xpbd_world_create",
    "xpbd_world_destroy", "xpbd_world_set_shapes", "xpbd_world_upload_bodies", "xpbd_world_download_bodies",
    "xpbd_world_body_count", "xpbd_world_download_frames", "xpbd_world_step", "xpbd_world_synchronize", "xpbd_world_download_contacts",
    "xpbd_world_download_contact_masks", "xpbd_world_set_stream", "xpbd_world_get_stream", "xpbd_world_set_mode",
    "xpbd_step_one", "xpbd_selftest_div_sqrt", "xpbd_world_set_polytopes", "xpbd_world_narrowphase",
    "xpbd_world_set_contact_pad", "xpbd_world_set_max_depenetration_speed", "xpbd_multi_world_set_max_depenetration_speed",
    "xpbd_world_contact_stats", "xpbd_world_build_neighbours",
    "xpbd_world_download_neighbours", "xpbd_world_contacts_begin", "xpbd_world_contacts_substep",
    "xpbd_world_export_dynamic", "xpbd_world_import_dynamic", "xpbd_world_import_dynamic_rows", "xpbd_world_set_joints",
    "xpbd_world_narrowphase_gjk", "xpbd_world_set_narrowphase",
    "xpbd_world_set_sat_schedule",
    "xpbd_world_edge_axes_separation",
    "xpbd_selftest_hbm_copy", "xpbd_selftest_field_streams", "xpbd_selftest_gather", "xpbd_world_snapshot_positions", "xpbd_world_max_displacement2",
    "xpbd_comm_unique_id", "xpbd_comm_library", "xpbd_multi_config_default", "xpbd_multi_world_create", "xpbd_multi_world_destroy",
    "xpbd_multi_world_set_polytopes", "xpbd_multi_world_upload", "xpbd_multi_world_step", "xpbd_multi_world_replan",
    "xpbd_multi_world_synchronize", "xpbd_multi_world_download", "xpbd_multi_world_halo_stats", "xpbd_multi_world_contact_stats",
    "xpbd_halo_cell_key", "xpbd_halo_plan", "xpbd_halo_plan_far", "xpbd_halo_partition", "xpbd_halo_plan_owned", "xpbd_halo_plan_light",
    "xpbd_multi_world_download_owned", "xpbd_multi_world_plan_stats", "xpbd_multi_world_owners",
    "xpbd_world_history_push", "xpbd_world_history_restore", "xpbd_world_history_truncate", "xpbd_world_history_length",
]


class XpbdError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("xpbd error %d: %s" % (code, msg))
        self.code = code


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("device", C.c_int32), ("mode", C.c_uint32), ("flags", C.c_uint32),
                ("block_size", C.c_uint32), ("reserved", C.c_uint32 * 3)]


_u32p = C.POINTER(C.c_uint32)
_f64p = C.POINTER(C.c_double)


class MultiConfig(C.Structure):
    """xpbd_multi_config"""
    _fields_ = [("struct_size", C.c_uint32), ("n_ranks", C.c_uint32), ("first_rank", C.c_uint32), ("n_local", C.c_uint32),
                ("devices", C.POINTER(C.c_int32)), ("transport", C.c_uint32), ("flags", C.c_uint32), ("comm_id", C.c_char_p),
                ("contact_pad", C.c_double), ("halo_margin", C.c_double), ("narrowphase", C.c_uint32), ("reserved", C.c_uint32)]


COMM_ID_BYTES = 128
TRANSPORT_RCCL, TRANSPORT_LOCAL = 0, 1
MULTI_AUTO_REPLAN = 1
MULTI_PLAN_THROUGH_DEVICE = 2
MULTI_SERIAL_ENQUEUE = 4
MULTI_FULL_PLANS = 8


class PolytopeDesc(C.Structure):
    """xpbd_polytope"""
    _fields_ = [("vertices_xyz", _f64p), ("edges", _u32p), ("face_offsets", _u32p), ("face_indices", _u32p),
                ("n_vertices", C.c_uint32), ("n_edges", C.c_uint32), ("n_faces", C.c_uint32), ("reserved", C.c_uint32),
                ("centroid", C.c_double * 3)]


# xpbd_joint as a numpy record (120 bytes)
JOINT_DTYPE = np.dtype([("body_a", "<u4"), ("body_b", "<u4"), ("anchor_a", "<f8", (3,)), ("anchor_b", "<f8", (3,)),
                        ("distance", "<f8"), ("axis_a", "<f8", (3,)), ("axis_b", "<f8", (3,)), ("kind", "<u4"), ("reserved", "<u4")])
JOINT_DISTANCE, JOINT_HINGE = 0, 1
# xpbd_gjk_result as a numpy record (96 bytes)
GJK_DTYPE = np.dtype([("status", "<i4"), ("gjk_iterations", "<u4"), ("epa_iterations", "<u4"), ("reserved", "<u4"),
                      ("depth", "<f8"), ("normal", "<f8", (3,)), ("point_a", "<f8", (3,)), ("point_b", "<f8", (3,))])
GJK_SEPARATED, GJK_PENETRATING, GJK_DEGENERATE = 0, 1, 2
# xpbd_edge_query as a numpy record (16 bytes)
EDGE_QUERY_DTYPE = np.dtype([("separation", "<f8"), ("edge_a", "<u4"), ("edge_b", "<u4")])
NARROWPHASE_SAT, NARROWPHASE_GJK_EPA = 0, 1
MAX_MANIFOLD_POINTS = 8
FEATURE_FACE_A, FEATURE_FACE_B, FEATURE_EDGES = 0, 1, 2
# xpbd_manifold as a numpy record (408 bytes)
MANIFOLD_DTYPE = np.dtype([("n_points", "<u4"), ("feature", "<u4"), ("index_a", "<u4"), ("index_b", "<u4"),
                           ("separation", "<f8"), ("p_ref", "<f8", (8, 3)), ("p_inc", "<f8", (8, 3))])


def _load(name):
    path = os.path.join(LIB_DIR, name)
    if name == "libxpbd_hip.so" and os.environ.get("XPBD_HIP_LIB"):     # A/B measurements of kernel variants (scripts/)
        path = os.environ["XPBD_HIP_LIB"]
    if not os.path.exists(path):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(no CPU fallback exists)" % path)
    return C.CDLL(path)


_hip = None
_host = None


def hip_lib():
    """libxpbd_hip.so with argtypes set.  Loading works without a GPU; compute calls then fail loudly."""
    global _hip
    if _hip is None:
        L = _load("libxpbd_hip.so")
        L.xpbd_abi_version.restype = C.c_uint32
        L.xpbd_last_error.restype = C.c_char_p
        L.xpbd_config_default.argtypes = [C.POINTER(Config)]
        L.xpbd_config_default.restype = None
        L.xpbd_world_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(Config)]
        L.xpbd_world_destroy.argtypes = [C.c_void_p]
        L.xpbd_world_destroy.restype = None
        L.xpbd_world_set_shapes.argtypes = [C.c_void_p, _f64p, _u32p, C.c_uint32]
        L.xpbd_world_upload_bodies.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.c_uint32]
        L.xpbd_world_download_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.xpbd_world_download_frames.argtypes = [C.c_void_p, _f64p, C.c_uint32]
        L.xpbd_world_body_count.argtypes = [C.c_void_p]
        L.xpbd_world_body_count.restype = C.c_uint32
        L.xpbd_world_step.argtypes = [C.c_void_p, C.c_double, C.c_uint32]
        L.xpbd_world_synchronize.argtypes = [C.c_void_p]
        L.xpbd_world_download_contacts.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, _u32p]
        L.xpbd_world_download_contact_masks.argtypes = [C.c_void_p, _u32p, C.c_uint32, C.c_uint32]
        L.xpbd_world_set_stream.argtypes = [C.c_void_p, C.c_void_p]
        L.xpbd_world_get_stream.argtypes = [C.c_void_p]
        L.xpbd_world_get_stream.restype = C.c_void_p
        L.xpbd_world_set_mode.argtypes = [C.c_void_p, C.c_uint32]
        L.xpbd_step_one.argtypes = [C.c_void_p, _f64p, C.c_uint32, C.c_double, C.c_uint32]
        L.xpbd_selftest_div_sqrt.argtypes = [C.c_int32, _f64p, _f64p, _f64p, _f64p, C.c_uint32]
        L.xpbd_selftest_hbm_copy.argtypes = [C.c_int32, C.c_uint64, C.c_uint32, _f64p]
        L.xpbd_selftest_field_streams.argtypes = [C.c_int32, C.c_uint64, C.c_uint32, C.c_uint32, _f64p]
        L.xpbd_world_snapshot_positions.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.xpbd_world_max_displacement2.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
        L.xpbd_comm_unique_id.argtypes = [C.c_char_p]
        L.xpbd_multi_config_default.argtypes = [C.POINTER(MultiConfig)]
        L.xpbd_multi_config_default.restype = None
        L.xpbd_multi_world_create.argtypes = [C.POINTER(C.c_void_p), C.POINTER(MultiConfig)]
        L.xpbd_multi_world_destroy.argtypes = [C.c_void_p]
        L.xpbd_multi_world_destroy.restype = None
        L.xpbd_multi_world_set_polytopes.argtypes = [C.c_void_p, C.POINTER(PolytopeDesc), C.c_uint32]
        L.xpbd_multi_world_upload.argtypes = [C.c_void_p, C.c_void_p, _u32p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32]
        L.xpbd_multi_world_step.argtypes = [C.c_void_p, C.c_double, C.c_uint32]
        L.xpbd_multi_world_replan.argtypes = [C.c_void_p]
        L.xpbd_multi_world_synchronize.argtypes = [C.c_void_p]
        L.xpbd_multi_world_download.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.xpbd_multi_world_halo_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), _f64p]
        L.xpbd_multi_world_contact_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        try:
            L.xpbd_multi_world_download_owned.argtypes = [C.c_void_p, _u32p, C.c_void_p, C.c_uint32, _u32p]
            L.xpbd_multi_world_plan_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
            L.xpbd_multi_world_owners.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        except AttributeError:          # an older build loaded through XPBD_HIP_LIB for an A/B measurement (scripts/)
            pass
        L.xpbd_halo_cell_key.argtypes = [_f64p, C.c_double]
        L.xpbd_halo_cell_key.restype = C.c_int64
        L.xpbd_halo_plan.argtypes = [C.POINTER(C.c_int64), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p,
                                     _u32p, C.c_uint32]
        L.xpbd_world_set_polytopes.argtypes = [C.c_void_p, C.POINTER(PolytopeDesc), C.c_uint32]
        L.xpbd_world_narrowphase.argtypes = [C.c_void_p, _u32p, C.c_uint32, C.c_void_p]
        L.xpbd_world_edge_axes_separation.argtypes = [C.c_void_p, _u32p, C.c_uint32, C.c_void_p]
        L.xpbd_world_narrowphase_gjk.argtypes = [C.c_void_p, _u32p, C.c_uint32, C.c_void_p]
        L.xpbd_world_set_narrowphase.argtypes = [C.c_void_p, C.c_uint32]
        L.xpbd_world_set_contact_pad.argtypes = [C.c_void_p, C.c_double]
        try:
            L.xpbd_world_set_max_depenetration_speed.argtypes = [C.c_void_p, C.c_double]
            L.xpbd_multi_world_set_max_depenetration_speed.argtypes = [C.c_void_p, C.c_double]
        except AttributeError:          # an older build loaded through XPBD_HIP_LIB
            pass
        L.xpbd_world_contact_stats.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.xpbd_world_build_neighbours.argtypes = [C.c_void_p, C.c_double, _u32p]
        L.xpbd_world_download_neighbours.argtypes = [C.c_void_p, _u32p, _u32p, C.c_uint32]
        L.xpbd_world_set_joints.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.xpbd_world_contacts_begin.argtypes = [C.c_void_p, C.c_double]
        L.xpbd_world_contacts_substep.argtypes = [C.c_void_p, C.c_double]
        L.xpbd_world_export_dynamic.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.xpbd_world_import_dynamic.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.xpbd_world_import_dynamic_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p]
        L.xpbd_world_set_sat_schedule.argtypes = [C.c_void_p, C.c_uint32]
        L.xpbd_world_history_push.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.xpbd_world_history_restore.argtypes = [C.c_void_p, C.c_uint32]
        L.xpbd_world_history_truncate.argtypes = [C.c_void_p, C.c_uint32]
        L.xpbd_world_history_length.argtypes = [C.c_void_p]
        L.xpbd_world_history_length.restype = C.c_uint32
        _hip = L
    return _hip


def host_lib():
    global _host
    if _host is None:
        hip_lib()  # libxpbd_host.so links against it
        L = _load("libxpbd_host.so")
        L.xpbdh_scene_shapes.argtypes = [C.c_uint32, _f64p, C.c_uint32, _u32p, C.c_uint32]
        L.xpbdh_scene_generate.argtypes = [C.c_uint32, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, _u32p]
        L.xpbdh_default_grid_width.argtypes = [C.c_uint32]
        L.xpbdh_default_grid_width.restype = C.c_uint32
        L.xpbdh_rigid_metrics.argtypes = [C.c_uint32, C.c_double, C.c_double, _f64p]
        L.xpbdh_rigid_metrics.restype = None
        L.xpbdh_rigid_new.argtypes = [_f64p, C.c_void_p]
        L.xpbdh_rigid_frame.argtypes = [C.c_void_p, _f64p]
        L.xpbdh_rigid_frame.restype = None
        L.xpbdh_world_new.argtypes = [C.c_void_p, C.c_void_p]
        L.xpbdh_shape_plane.argtypes = [C.c_uint32, C.c_double, C.c_uint32, _f64p]
        L.xpbdh_polytope_arrays.argtypes = [C.c_uint32, C.c_double, _u32p, _f64p, _u32p, _u32p, _u32p, _f64p]
        L.xpbdh_polytope_arrays.restype = None
        _host = L
    return _host


def _check(rc):
    if rc != OK:
        raise XpbdError(rc, hip_lib().xpbd_last_error().decode())


def _f64(a):
    return a.ctypes.data_as(_f64p)


def _u32(a):
    return a.ctypes.data_as(_u32p)


class World:
    """N-body world on one GPU: thin wrapper over xpbd_world_* (reference World::integrate, src/world.rs:34-43)."""

    def __init__(self, device=0, mode=MODE_FUSED, trace_contacts=False, block_size=0):
        L = hip_lib()
        cfg = Config()
        L.xpbd_config_default(C.byref(cfg))
        cfg.device, cfg.mode, cfg.block_size = device, mode, block_size
        cfg.flags = FLAG_TRACE_CONTACTS if trace_contacts else 0
        self._h = C.c_void_p()
        _check(L.xpbd_world_create(C.byref(self._h), C.byref(cfg)))
        self.n = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            hip_lib().xpbd_world_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_shapes(self, verts_xyz, vert_offsets):
        v = np.ascontiguousarray(verts_xyz, dtype=np.float64).reshape(-1)
        o = np.ascontiguousarray(vert_offsets, dtype=np.uint32)
        if v.size == 0:
            v = np.zeros(3)
        _check(hip_lib().xpbd_world_set_shapes(self._h, _f64(v), _u32(o), o.size - 1))

    def upload(self, bodies, shape_id=None):
        b = np.ascontiguousarray(bodies, dtype=np.float64).reshape(-1, RIGID_DOUBLES)
        sid = None if shape_id is None else np.ascontiguousarray(shape_id, dtype=np.uint32)
        _check(hip_lib().xpbd_world_upload_bodies(self._h, b.ctypes.data, None if sid is None else _u32(sid), b.shape[0]))
        self.n = b.shape[0]

    def step(self, dt, substeps):
        _check(hip_lib().xpbd_world_step(self._h, dt, substeps))

    def synchronize(self):
        _check(hip_lib().xpbd_world_synchronize(self._h))

    def download(self):
        out = np.empty((self.n, RIGID_DOUBLES), dtype=np.float64)
        _check(hip_lib().xpbd_world_download_bodies(self._h, out.ctypes.data, self.n))
        return out

    def frames(self):
        """(n, 7) Rigid::frame() of every body: origin xyz, rotation sxyz (what the reference's renderer reads)."""
        out = np.empty((self.n, 7), dtype=np.float64)
        _check(hip_lib().xpbd_world_download_frames(self._h, _f64(out), self.n))
        return out

    def contacts(self):
        """(k, 2) uint32 array of (body, vertex) of the last substep, reference push order."""
        n = C.c_uint32(0)
        rc = hip_lib().xpbd_world_download_contacts(self._h, None, 0, C.byref(n))
        if rc not in (OK, E_CAPACITY):
            _check(rc)
        out = np.empty((n.value, 2), dtype=np.uint32)
        if n.value:
            _check(hip_lib().xpbd_world_download_contacts(self._h, out.ctypes.data, n.value, C.byref(n)))
        return out

    def contact_masks(self, substeps):
        out = np.empty((substeps, self.n), dtype=np.uint32)
        _check(hip_lib().xpbd_world_download_contact_masks(self._h, _u32(out), substeps, self.n))
        return out

    def set_polytopes(self, polytopes):
        """polytopes: list of dicts with vertices (V,3), edges (E,2), face_offsets (F+1), face_indices, centroid (3)."""
        descs, _keep = polytope_descs(polytopes)
        _check(hip_lib().xpbd_world_set_polytopes(self._h, descs, len(polytopes)))

    def narrowphase(self, pairs):
        """SAT manifolds (MANIFOLD_DTYPE records) of the given (A, B) body pairs at the current poses."""
        pr = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros(pr.shape[0], dtype=MANIFOLD_DTYPE)
        _check(hip_lib().xpbd_world_narrowphase(self._h, _u32(pr), pr.shape[0], out.ctypes.data))
        return out

    def edge_axes_separation(self, pairs):
        """The reference's edge_axes_separation (src/collision.rs:151-197) of the given (A, B) pairs: EDGE_QUERY_DTYPE records."""
        pr = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros(pr.shape[0], dtype=EDGE_QUERY_DTYPE)
        _check(hip_lib().xpbd_world_edge_axes_separation(self._h, _u32(pr), pr.shape[0], out.ctypes.data))
        return out

    def narrowphase_gjk(self, pairs):
        """GJK + EPA results (GJK_DTYPE records) of the given (A, B) body pairs at the current poses."""
        pr = np.ascontiguousarray(pairs, dtype=np.uint32).reshape(-1, 2)
        out = np.zeros(pr.shape[0], dtype=GJK_DTYPE)
        _check(hip_lib().xpbd_world_narrowphase_gjk(self._h, _u32(pr), pr.shape[0], out.ctypes.data))
        return out

    def set_narrowphase(self, narrowphase):
        _check(hip_lib().xpbd_world_set_narrowphase(self._h, narrowphase))

    def set_contact_pad(self, pad):
        _check(hip_lib().xpbd_world_set_contact_pad(self._h, pad))

    def set_max_depenetration_speed(self, speed):
        """Limit on how fast a body-body contact may push its bodies apart (m/s); 0 = off = the reference's solver loop."""
        _check(hip_lib().xpbd_world_set_max_depenetration_speed(self._h, speed))

    def contact_stats(self):
        """(neighbour pairs of the last step, touching pairs, manifold points) -- the last two since the previous call."""
        out = (C.c_uint64 * 3)()
        _check(hip_lib().xpbd_world_contact_stats(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])

    def neighbours(self, dt):
        """Runs the sphere broadphase as step(dt, .) would: (offsets[n+1], neighbours) CSR, ascending."""
        n_entries = C.c_uint32(0)
        _check(hip_lib().xpbd_world_build_neighbours(self._h, dt, C.byref(n_entries)))
        off = np.zeros(self.n + 1, dtype=np.uint32)
        nb = np.zeros(max(n_entries.value, 1), dtype=np.uint32)
        _check(hip_lib().xpbd_world_download_neighbours(self._h, _u32(off), _u32(nb), nb.size))
        return off, nb[: n_entries.value]

    def set_joints(self, joints):
        """joints: JOINT_DTYPE records naming bodies of the last upload (extension; XPBD_MODE_CONTACTS)."""
        j = np.ascontiguousarray(joints, dtype=JOINT_DTYPE)
        _check(hip_lib().xpbd_world_set_joints(self._h, j.ctypes.data if j.size else None, j.size))

    def contacts_begin(self, dt):
        _check(hip_lib().xpbd_world_contacts_begin(self._h, dt))

    def contacts_substep(self, h):
        _check(hip_lib().xpbd_world_contacts_substep(self._h, h))

    def export_dynamic(self, dev_indices_ptr, n, dev_buf_ptr):
        """Device pointers (e.g. torch tensor .data_ptr()): n uint32 indices, n x 13 doubles."""
        _check(hip_lib().xpbd_world_export_dynamic(self._h, C.c_void_p(dev_indices_ptr), n, C.c_void_p(dev_buf_ptr)))

    def import_dynamic(self, dev_indices_ptr, n, dev_buf_ptr):
        _check(hip_lib().xpbd_world_import_dynamic(self._h, C.c_void_p(dev_indices_ptr), n, C.c_void_p(dev_buf_ptr)))

    def import_dynamic_rows(self, dev_indices_ptr, dev_rows_ptr, n, dev_buf_ptr):
        """Body indices[k] takes row rows[k] of the buffer (device pointers; rows are uint32)."""
        _check(hip_lib().xpbd_world_import_dynamic_rows(self._h, C.c_void_p(dev_indices_ptr), C.c_void_p(dev_rows_ptr), n,
                                                        C.c_void_p(dev_buf_ptr)))

    def set_sat_schedule(self, schedule):
        """SAT_SCHEDULE_AUTO / _ONE_PASS / _TWO_PASS (same results, different cost)."""
        _check(hip_lib().xpbd_world_set_sat_schedule(self._h, schedule))

    # state history: the reference app's `states` vector and `current_state` cursor (src/app.rs:48, 206-212)
    def history_push(self):
        index = C.c_uint32(0)
        _check(hip_lib().xpbd_world_history_push(self._h, C.byref(index)))
        return index.value

    def history_restore(self, index):
        _check(hip_lib().xpbd_world_history_restore(self._h, index))

    def history_truncate(self, length):
        _check(hip_lib().xpbd_world_history_truncate(self._h, length))

    def history_length(self):
        return hip_lib().xpbd_world_history_length(self._h)

    def set_stream(self, stream_ptr):
        _check(hip_lib().xpbd_world_set_stream(self._h, C.c_void_p(stream_ptr)))

    def set_mode(self, mode):
        _check(hip_lib().xpbd_world_set_mode(self._h, mode))


def polytope_descs(polytopes):
    """(xpbd_polytope array, the numpy arrays it points into) from a list of polytope dicts."""
    keep, descs = [], (PolytopeDesc * len(polytopes))()
    for d, p in zip(descs, polytopes):
        v = np.ascontiguousarray(p["vertices"], dtype=np.float64).reshape(-1, 3)
        e = np.ascontiguousarray(p["edges"], dtype=np.uint32).reshape(-1, 2)
        fo = np.ascontiguousarray(p["face_offsets"], dtype=np.uint32)
        fi = np.ascontiguousarray(p["face_indices"], dtype=np.uint32)
        keep += [v, e, fo, fi]
        d.vertices_xyz, d.edges, d.face_offsets, d.face_indices = _f64(v), _u32(e), _u32(fo), _u32(fi)
        d.n_vertices, d.n_edges, d.n_faces = v.shape[0], e.shape[0], fo.size - 1
        d.centroid[:] = [float(x) for x in p["centroid"]]
    return descs, keep


def pile_depth(count, pitch, layers):
    """Extent in y (metres, one pitch of clearance included) of a scene_pile of `count` bodies: piles laid side by side at
    this spacing form one continuous pile."""
    per_layer = (count + layers - 1) // layers
    w = default_grid_width(per_layer)
    return pitch * ((per_layer + w - 1) // w)


def comm_library():
    """Path of the RCCL library the multi-GPU world is bound to (None if none could be loaded)."""
    L = hip_lib()
    L.xpbd_comm_library.restype = C.c_char_p
    p = L.xpbd_comm_library()
    return p.decode() if p else None


def comm_unique_id():
    """XPBD_COMM_ID_BYTES of a fresh RCCL communicator id (made on one rank, handed to all)."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    _check(hip_lib().xpbd_comm_unique_id(buf))
    return buf.raw


class MultiWorld:
    """xpbd_multi_world: this process's local shards of an N-body world with body-body contacts sharded over n_ranks GPUs
    (EXTENSION; the caller is World::integrate, src/world.rs:34-43)."""

    def __init__(self, n_ranks, first_rank=0, devices=(0,), transport=TRANSPORT_RCCL, comm_id=None, pad=0.02, halo_margin=0.5,
                 narrowphase=NARROWPHASE_SAT, auto_replan=False, plan_through_device=False, serial_enqueue=False, full_plans=False):
        L = hip_lib()
        cfg = MultiConfig()
        L.xpbd_multi_config_default(C.byref(cfg))
        self._devices = (C.c_int32 * len(devices))(*devices)
        self._comm_id = comm_id
        cfg.n_ranks, cfg.first_rank, cfg.n_local, cfg.devices = n_ranks, first_rank, len(devices), self._devices
        cfg.transport, cfg.comm_id = transport, comm_id
        cfg.flags = (MULTI_AUTO_REPLAN if auto_replan else 0) | (MULTI_PLAN_THROUGH_DEVICE if plan_through_device else 0) \
            | (MULTI_SERIAL_ENQUEUE if serial_enqueue else 0) | (MULTI_FULL_PLANS if full_plans else 0)
        cfg.contact_pad, cfg.halo_margin, cfg.narrowphase = pad, halo_margin, narrowphase
        self._h = C.c_void_p()
        _check(L.xpbd_multi_world_create(C.byref(self._h), C.byref(cfg)))
        self.n = self.n_global = 0

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            hip_lib().xpbd_multi_world_destroy(self._h)
            self._h = C.c_void_p()

    __del__ = close

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def set_polytopes(self, polytopes):
        descs, _keep = polytope_descs(polytopes)
        _check(hip_lib().xpbd_multi_world_set_polytopes(self._h, descs, len(polytopes)))

    def set_max_depenetration_speed(self, speed):
        _check(hip_lib().xpbd_multi_world_set_max_depenetration_speed(self._h, speed))

    def upload(self, bodies, shape_id, first_global, n_global, joints=None):
        """bodies / shape_id: the slice of the caller's bodies this process hands over, global indices [first_global,
        first_global + len), in any order: the library decides who owns what."""
        b = np.ascontiguousarray(bodies, dtype=np.float64).reshape(-1, RIGID_DOUBLES)
        sid = None if shape_id is None else np.ascontiguousarray(shape_id, dtype=np.uint32)
        j = np.zeros(0, dtype=JOINT_DTYPE) if joints is None else np.ascontiguousarray(joints, dtype=JOINT_DTYPE)
        _check(hip_lib().xpbd_multi_world_upload(self._h, b.ctypes.data, None if sid is None else _u32(sid), first_global, b.shape[0],
                                                 n_global, j.ctypes.data if j.size else None, j.size))
        self.n, self.n_global = b.shape[0], n_global

    def step(self, dt, substeps):
        _check(hip_lib().xpbd_multi_world_step(self._h, dt, substeps))

    def replan(self):
        _check(hip_lib().xpbd_multi_world_replan(self._h))

    def synchronize(self):
        _check(hip_lib().xpbd_multi_world_synchronize(self._h))

    def download(self):
        out = np.empty((self.n, RIGID_DOUBLES), dtype=np.float64)
        _check(hip_lib().xpbd_multi_world_download(self._h, out.ctypes.data, self.n))
        return out

    def download_owned(self):
        """(global ids, (k, 38) states) of the bodies this process's shards own at the moment.  Not collective."""
        n = C.c_uint32(0)
        cap = max(self.n_global, 1)
        ids, out = np.empty(cap, dtype=np.uint32), np.empty((cap, RIGID_DOUBLES), dtype=np.float64)
        _check(hip_lib().xpbd_multi_world_download_owned(self._h, _u32(ids), out.ctypes.data, cap, C.byref(n)))
        return ids[: n.value].copy(), out[: n.value].copy()

    def plan_stats(self):
        """dict: plans, frames undone, bodies that changed owner at the last plan, fewest / most bodies owned by a rank, step
        calls and the host nanoseconds inside them (enqueueing, waiting for the pair counts, waiting for the frame's end), and the
        host nanoseconds spent in plans (creation, re-plans)."""
        out = (C.c_uint64 * 12)()
        _check(hip_lib().xpbd_multi_world_plan_stats(self._h, out))
        keys = ("plans", "rollbacks", "migrated", "owned_min", "owned_max", "steps", "ns_enqueue", "ns_wait_broadphase", "ns_wait_frame", "ns_plan",
                "full_plans", "light_plans")
        return dict(zip(keys, (int(x) for x in out)))

    def owners(self):
        out = np.empty(self.n_global, dtype=np.uint8)
        _check(hip_lib().xpbd_multi_world_owners(self._h, out.ctypes.data, self.n_global))
        return out

    def halo_stats(self):
        """dict: bodies of the world, owned / ghost / boundary bodies here, rows per rank of the all-gather, plans made, and
        the largest distance a body had travelled at the last check."""
        out, moved = (C.c_uint64 * 6)(), C.c_double(0.0)
        _check(hip_lib().xpbd_multi_world_halo_stats(self._h, out, C.byref(moved)))
        keys = ("bodies_global", "owned", "ghosts", "boundary", "rows_per_rank", "plans")
        return dict(zip(keys, (int(x) for x in out)), max_displacement=moved.value)

    def contact_stats(self):
        out = (C.c_uint64 * 3)()
        _check(hip_lib().xpbd_multi_world_contact_stats(self._h, out))
        return int(out[0]), int(out[1]), int(out[2])


def halo_cell_key(centre, edge):
    c = np.ascontiguousarray(centre, dtype=np.float64)
    return int(hip_lib().xpbd_halo_cell_key(_f64(c), edge))


def halo_plan(cell_keys, n_ranks, rank, joints=None):
    """(ghost ids, boundary ids) of one rank from the grid-cell keys of ALL bodies: host-only, as xpbd_multi_world_upload plans."""
    keys = np.ascontiguousarray(cell_keys, dtype=np.int64)
    j = np.zeros(0, dtype=JOINT_DTYPE) if joints is None else np.ascontiguousarray(joints, dtype=JOINT_DTYPE)
    ghosts, boundary = np.zeros(len(keys), dtype=np.uint32), np.zeros(len(keys), dtype=np.uint32)
    ng, nb = C.c_uint32(0), C.c_uint32(0)
    _check(hip_lib().xpbd_halo_plan(keys.ctypes.data_as(C.POINTER(C.c_int64)), len(keys), n_ranks, rank, j.ctypes.data if j.size else None,
                                    j.size, _u32(ghosts), C.byref(ng), _u32(boundary), C.byref(nb), len(keys)))
    return ghosts[: ng.value].copy(), boundary[: nb.value].copy()


def halo_partition(cell_keys, n_ranks):
    """owner[g] of every body from the grid-cell keys of ALL bodies: host-only, as xpbd_multi_world_upload cuts the shards."""
    keys = np.ascontiguousarray(cell_keys, dtype=np.int64)
    owner = np.zeros(len(keys), dtype=np.uint8)
    L = hip_lib()
    L.xpbd_halo_partition.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    _check(L.xpbd_halo_partition(keys.ctypes.data, len(keys), n_ranks, owner.ctypes.data))
    return owner


def halo_plan_owned(cell_keys, owner, n_ranks, rank, joints=None, with_far=False):
    """(ghost ids, boundary ids[, far flags per owned body]) of one rank from the cell keys and owners of ALL bodies."""
    keys = np.ascontiguousarray(cell_keys, dtype=np.int64)
    own = np.ascontiguousarray(owner, dtype=np.uint8)
    j = np.zeros(0, dtype=JOINT_DTYPE) if joints is None else np.ascontiguousarray(joints, dtype=JOINT_DTYPE)
    ghosts, boundary = np.zeros(len(keys), dtype=np.uint32), np.zeros(len(keys), dtype=np.uint32)
    far = np.zeros(len(keys), dtype=np.uint8)
    ng, nb = C.c_uint32(0), C.c_uint32(0)
    L = hip_lib()
    L.xpbd_halo_plan_owned.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, _u32p, _u32p, _u32p,
                                       _u32p, C.c_void_p, C.c_uint32]
    _check(L.xpbd_halo_plan_owned(keys.ctypes.data, own.ctypes.data, len(keys), n_ranks, rank, j.ctypes.data if j.size else None, j.size,
                                  _u32(ghosts), C.byref(ng), _u32(boundary), C.byref(nb), far.ctypes.data if with_far else None, len(keys)))
    if with_far:
        return ghosts[: ng.value].copy(), boundary[: nb.value].copy(), far[: int((own == rank).sum())].copy()
    return ghosts[: ng.value].copy(), boundary[: nb.value].copy()


def halo_plan_light(keys_at_cut, cell_keys, n_ranks, rank, joints=None):
    """(owner of every body now, own ids, ghost ids, boundary ids, far flags) of one rank's LIGHT plan (host only): cuts from
    `keys_at_cut`, owners from `cell_keys` and those cuts, halos from the rank's own bodies and everybody's rims."""
    k0 = np.ascontiguousarray(keys_at_cut, dtype=np.int64)
    k1 = np.ascontiguousarray(cell_keys, dtype=np.int64)
    n = len(k1)
    j = np.zeros(0, dtype=JOINT_DTYPE) if joints is None else np.ascontiguousarray(joints, dtype=JOINT_DTYPE)
    owner = np.zeros(n, dtype=np.uint8)
    own, ghosts, boundary = (np.zeros(n, dtype=np.uint32) for _ in range(3))
    far = np.zeros(n, dtype=np.uint8)
    no, ng, nb = C.c_uint32(0), C.c_uint32(0), C.c_uint32(0)
    L = hip_lib()
    L.xpbd_halo_plan_light.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, C.c_void_p, _u32p, _u32p,
                                       _u32p, _u32p, _u32p, _u32p, C.c_void_p, C.c_uint32]
    _check(L.xpbd_halo_plan_light(k0.ctypes.data, k1.ctypes.data, n, n_ranks, rank, j.ctypes.data if j.size else None, j.size, owner.ctypes.data,
                                  _u32(own), C.byref(no), _u32(ghosts), C.byref(ng), _u32(boundary), C.byref(nb), far.ctypes.data, n))
    return owner, own[: no.value].copy(), ghosts[: ng.value].copy(), boundary[: nb.value].copy(), far[: no.value].copy()


def halo_plan_far(cell_keys, n_ranks, rank):
    """Per owned body of `rank` (index order): 1 if the plan classes it as far from every foreign body (host-only)."""
    keys = np.ascontiguousarray(cell_keys, dtype=np.int64)
    far = np.zeros(len(keys), dtype=np.uint8)
    n = C.c_uint32(0)
    L = hip_lib()
    L.xpbd_halo_plan_far.argtypes = [C.POINTER(C.c_int64), C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_uint32, _u32p]
    _check(L.xpbd_halo_plan_far(keys.ctypes.data_as(C.POINTER(C.c_int64)), len(keys), n_ranks, rank, far.ctypes.data, len(keys), C.byref(n)))
    return far[: n.value].copy()


def step_one(rigid, verts_xyz, dt, substeps):
    """solver::step for one body (src/solver.rs:3) through xpbd_step_one; returns the new 38-double state."""
    r = np.array(rigid, dtype=np.float64).reshape(RIGID_DOUBLES).copy()
    v = np.ascontiguousarray(verts_xyz, dtype=np.float64).reshape(-1)
    _check(hip_lib().xpbd_step_one(r.ctypes.data, _f64(v), v.size // 3, dt, substeps))
    return r


def selftest_div_sqrt(a, b, device=0):
    """(a / b, sqrt(a)) computed on the GPU with the kernels' own compile flags."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = np.ascontiguousarray(b, dtype=np.float64)
    q, s = np.empty_like(a), np.empty_like(a)
    _check(hip_lib().xpbd_selftest_div_sqrt(device, _f64(a), _f64(b), _f64(q), _f64(s), a.size))
    return q, s


def selftest_gather(records, record_bytes, read_bytes, repeats=5, device=0):
    """GB/s of a gather of known size (every record read once, 16-byte loads): the FETCH_SIZE calibration kernel."""
    out = C.c_double(0.0)
    L = hip_lib()
    L.xpbd_selftest_gather.argtypes = [C.c_int32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, _f64p]
    _check(L.xpbd_selftest_gather(device, records, record_bytes, read_bytes, repeats, C.byref(out)))
    return out.value


def selftest_hbm_copy(nbytes=1 << 31, repeats=10, device=0):
    """GB/s (read + written) of a device-to-device copy by the library's streaming kernel: the measured HBM roof."""
    out = C.c_double(0.0)
    _check(hip_lib().xpbd_selftest_hbm_copy(device, nbytes, repeats, C.byref(out)))
    return out.value


def selftest_field_streams(bodies=2097152, tile_major=False, repeats=10, device=0):
    """GB/s of the access pattern of one substep of the pinned path alone (38 doubles in, 13 out per body): the roof of
    XPBD_MODE_PER_SUBSTEP in the world's field-major layout, or in a tile-major one."""
    out = C.c_double(0.0)
    _check(hip_lib().xpbd_selftest_field_streams(device, bodies, 1 if tile_major else 0, repeats, C.byref(out)))
    return out.value


# ---- host mirror (CPU set-up math; no GPU needed) -------------------------------
SAT_SCHEDULE_AUTO, SAT_SCHEDULE_ONE_PASS, SAT_SCHEDULE_TWO_PASS = 0, 1, 2
SCENE_BOXES, SCENE_MIXED, SCENE_BOXES_DROP, SCENE_MIXED_DROP, SCENE_BOX_STACKS = 0, 1, 2, 3, 4
SHAPE_CUBE, SHAPE_TETRAHEDRON, SHAPE_ICOSAHEDRON = 0, 1, 2


def scene_shapes(kind):
    verts = np.zeros(3 * 64)
    off = np.zeros(9, dtype=np.uint32)
    ns = host_lib().xpbdh_scene_shapes(kind, _f64(verts), 64, _u32(off), 8)
    if ns < 0:
        raise XpbdError(E_CAPACITY, "scene shape tables too large")
    off = off[: ns + 1].copy()
    return verts[: 3 * off[-1]].reshape(-1, 3).copy(), off


def default_grid_width(n):
    return int(host_lib().xpbdh_default_grid_width(n))


def scene_generate(kind, seed, n, first=0, count=None, grid_w=None):
    """Bodies [first, first+count) of the n-body seeded scene: ((count,38) f64, (count,) u32)."""
    count = n - first if count is None else count
    grid_w = default_grid_width(n) if grid_w is None else grid_w
    bodies = np.zeros((count, RIGID_DOUBLES))
    sid = np.zeros(count, dtype=np.uint32)
    rc = host_lib().xpbdh_scene_generate(kind, seed, grid_w, first, count, bodies.ctypes.data, _u32(sid))
    if rc != OK:
        raise XpbdError(rc, "scene generation failed")
    return bodies, sid


def scene_pile(kind, seed, n, pitch, layers, layer_gap=2.5, lift=0.6, first=0, count=None, y_offset=0.0):
    """The seeded scene re-gridded into a PILE for the body-body contact extension: `layers` layers of a square grid at
    `pitch` metres, `layer_gap` metres apart, the CENTRES OF MASS on the grid points (world centre = position +
    center_of_mass whatever the rotation, src/rigid.rs:75-80; the shapes' origins are a corner for the cube and the
    tetrahedron and the centre for the icosahedron), heights lifted by `lift`.  With the defaults nothing overlaps at
    t = 0: cube radius 0.866 about its centre, half-size tetrahedron 0.42, icosahedron 0.5; the generator's heights jitter by
    0.6 m; the mixed grid never puts two cubes side by side when the grid width is not a multiple of 3 (shape = index
    mod 3) and pitch >= 1.4 keeps a cube clear of the smaller shapes; boxes need pitch >= 1.75.  A scene with deep
    initial overlaps is resolved in ONE substep (XPBD has no velocity clamp), i.e. depth / h = 120 m/s for 0.1 m, and
    never reaches a steady contact regime."""
    count = n - first if count is None else count     # bodies [first, first + count) of the n-body seeded scene form the pile
    bodies, sid = scene_generate(kind, seed, n, first=first, count=count)
    per_layer = (count + layers - 1) // layers
    w = default_grid_width(per_layer)
    i = np.arange(count)
    k = i % per_layer
    bodies[:, 31] = pitch * (k % w) - bodies[:, 28]
    bodies[:, 32] = y_offset + pitch * (k // w) - bodies[:, 29]
    bodies[:, 33] += lift + layer_gap * (i // per_layer) + (0.5 - bodies[:, 30])
    return bodies, sid


def rigid_metrics(shape, scale, density):
    out = np.zeros(14)
    host_lib().xpbdh_rigid_metrics(shape, scale, density, _f64(out))
    return out


def rigid_new(metrics):
    out = np.zeros(RIGID_DOUBLES)
    m = np.ascontiguousarray(metrics, dtype=np.float64)
    rc = host_lib().xpbdh_rigid_new(_f64(m), out.ctypes.data)
    if rc != OK:
        raise XpbdError(rc, "Inertia tensor is not invertible")
    return out


def rigid_frame(rigid):
    r = np.ascontiguousarray(rigid, dtype=np.float64)
    out = np.zeros(7)
    host_lib().xpbdh_rigid_frame(r.ctypes.data, _f64(out))
    return out


def world_new():
    a, b = np.zeros(RIGID_DOUBLES), np.zeros(RIGID_DOUBLES)
    rc = host_lib().xpbdh_world_new(a.ctypes.data, b.ctypes.data)
    if rc != OK:
        raise XpbdError(rc, "World::new failed")
    return a, b


def polytope(shape, scale=1.0):
    """Topology of a standard shape (host mirror's Polytope) as the dict World.set_polytopes takes."""
    counts = np.zeros(4, dtype=np.uint32)
    host_lib().xpbdh_polytope_arrays(shape, scale, _u32(counts), None, None, None, None, None)
    v = np.zeros((counts[0], 3))
    e = np.zeros((counts[1], 2), dtype=np.uint32)
    fo = np.zeros(counts[2] + 1, dtype=np.uint32)
    fi = np.zeros(counts[3], dtype=np.uint32)
    c = np.zeros(3)
    host_lib().xpbdh_polytope_arrays(shape, scale, _u32(counts), _f64(v), _u32(e), _u32(fo), _u32(fi), _f64(c))
    return {"vertices": v, "edges": e, "face_offsets": fo, "face_indices": fi, "centroid": c}


def scene_polytopes(kind):
    """Polytopes of a scene kind, in shape-id order (matches scene_shapes)."""
    if not (kind & 1):
        return [polytope(SHAPE_CUBE)]
    return [polytope(SHAPE_CUBE), polytope(SHAPE_TETRAHEDRON, 0.5), polytope(SHAPE_ICOSAHEDRON, 0.5)]


def shape_planes(shape, scale=1.0):
    out = np.zeros(4)
    n = host_lib().xpbdh_shape_plane(shape, scale, 0xFFFFFFFF, _f64(out))
    planes = np.zeros((n, 4))
    for i in range(n):
        host_lib().xpbdh_shape_plane(shape, scale, i, _f64(planes[i]))
    return planes
