"""ctypes binding of oracle/libxpbd_oracle.so -- the CPU restatement of the reference.

Test infrastructure: imported only by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  Never by the product.
"""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "libxpbd_oracle.so")


class Vec3(C.Structure):
    _fields_ = [("x", C.c_double), ("y", C.c_double), ("z", C.c_double)]

    def np(self):
        return np.array([self.x, self.y, self.z])


class Quat(C.Structure):
    _fields_ = [("s", C.c_double), ("v", Vec3)]

    def np(self):
        return np.array([self.s, self.v.x, self.v.y, self.v.z])


class Mat3(C.Structure):
    _fields_ = [("x", Vec3), ("y", Vec3), ("z", Vec3)]

    def np(self):  # [col][row]
        return np.array([self.x.np(), self.y.np(), self.z.np()])


class Frame(C.Structure):
    _fields_ = [("position", Vec3), ("rotation", Quat)]


class Plane(C.Structure):
    _fields_ = [("normal", Vec3), ("displacement", C.c_double)]


class Rigid(C.Structure):
    _fields_ = [("inverse_mass", C.c_double), ("inverse_inertia", Mat3), ("external_force", Vec3),
                ("internal_force", Vec3), ("external_torque", Vec3), ("internal_torque", Vec3),
                ("velocity", Vec3), ("angular_velocity", Vec3), ("center_of_mass", Vec3), ("position", Vec3),
                ("rotation", Quat)]

    def np(self):
        return np.frombuffer(bytes(self), dtype=np.float64).copy()

    @classmethod
    def from_np(cls, a):
        a = np.ascontiguousarray(a, dtype=np.float64).reshape(38)
        return cls.from_buffer_copy(a.tobytes())


class Constraint(C.Structure):
    _fields_ = [("rigid", C.c_size_t), ("contact0", Vec3), ("contact1", Vec3), ("distance", C.c_double)]


class Metrics(C.Structure):
    _fields_ = [("mass", C.c_double), ("volume", C.c_double), ("center_of_mass", Vec3), ("inertia_tensor", Mat3)]


class Polytope(C.Structure):
    _fields_ = [("n_vertices", C.c_uint32), ("n_edges", C.c_uint32), ("n_faces", C.c_uint32),
                ("vertices", Vec3 * 32), ("edges", (C.c_uint32 * 2) * 64), ("face_offsets", C.c_uint32 * 33),
                ("face_indices", C.c_uint32 * 128), ("centroid", Vec3)]

    def verts(self):
        return np.array([self.vertices[i].np() for i in range(self.n_vertices)])


def vec3(a):
    return Vec3(float(a[0]), float(a[1]), float(a[2]))


def quat(a):
    return Quat(float(a[0]), vec3(a[1:4]))


def frame(pos, rot):
    return Frame(vec3(pos), quat(rot))


_lib = None


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO):
        raise ImportError("%s missing: run make -C oracle" % SO)
    L = C.CDLL(SO)
    P = C.POINTER
    sig = {
        "o_add": (Vec3, [Vec3, Vec3]), "o_sub": (Vec3, [Vec3, Vec3]), "o_dot": (C.c_double, [Vec3, Vec3]),
        "o_cross": (Vec3, [Vec3, Vec3]), "o_magnitude": (C.c_double, [Vec3]), "o_normalize": (Vec3, [Vec3]),
        "o_project_on": (Vec3, [Vec3, Vec3]), "o_qmul": (Quat, [Quat, Quat]), "o_qrot": (Vec3, [Quat, Vec3]),
        "o_qconj": (Quat, [Quat]), "o_qnormalize": (Quat, [Quat]),
        "o_quat_from_euler_deg": (Quat, [C.c_double] * 3), "o_mat3_mulv": (Vec3, [Mat3, Vec3]),
        "o_mat3_invert": (C.c_int, [Mat3, P(Mat3)]),
        "o_frame_inverse": (Frame, [Frame]), "o_frame_delta": (Vec3, [Frame, Frame, Vec3]),
        "o_frame_mulv": (Vec3, [Frame, Vec3]), "o_frame_mulplane": (Plane, [Frame, Plane]),
        "o_frame_mul": (Frame, [Frame, Frame]),
        "o_plane_from_points": (Plane, [Vec3, Vec3, Vec3]), "o_plane_from_point_normal": (Plane, [Vec3, Vec3]),
        "o_plane_distance": (C.c_double, [Plane, Vec3]),
        "o_polytope_tetrahedron": (None, [P(Polytope)]), "o_polytope_cube": (None, [P(Polytope)]),
        "o_polytope_icosahedron": (None, [P(Polytope)]), "o_polytope_scale": (None, [C.c_double, P(Polytope)]),
        "o_polytope_plane": (Plane, [P(Polytope), C.c_uint32]),
        "o_polytope_support": (Vec3, [P(Polytope), Frame, Vec3]),
        "o_polytope_minkowski_support": (Vec3, [P(Polytope), Frame, Frame, Vec3]),
        "o_rigid_metrics": (None, [P(Polytope), C.c_double, P(Metrics)]),
        "o_rigid_new": (C.c_int, [P(Metrics), P(Rigid)]), "o_rigid_frame": (Frame, [P(Rigid)]),
        "o_rigid_integrate": (None, [P(Rigid), C.c_double]),
        "o_rigid_derive": (None, [P(Rigid), Vec3, Quat, C.c_double]),
        "o_rigid_apply_impulse": (None, [P(Rigid), Vec3, Vec3]),
        "o_constraint_current_distance": (C.c_double, [P(Constraint)]),
        "o_constraint_inverse_resistance": (C.c_double, [P(Constraint), P(P(Rigid))]),
        "o_constraint_act": (None, [P(Constraint), P(P(Rigid)), C.c_double]),
        "o_ground": (C.c_uint32, [P(Rigid), Frame, P(Vec3), C.c_uint32, P(Constraint), P(C.c_uint32)]),
        "o_face_axes_separation": (C.c_double, [Frame, Frame, P(Polytope), P(Polytope), P(C.c_uint64)]),
        "o_edge_axes_separation": (C.c_double, [Frame, Frame, P(Polytope), P(Polytope), P(C.c_uint64), P(C.c_uint64)]),
        "o_solve": (None, [P(Rigid), P(Constraint), C.c_uint32, C.c_double]),
        "o_step": (None, [P(Rigid), P(Vec3), C.c_uint32, C.c_double, C.c_size_t, P(C.c_uint32)]),
        "o_world_new": (C.c_int, [P(Polytope), P(Polytope), P(Rigid), P(Rigid)]),
        "o_world_integrate": (None, [P(Rigid), P(Rigid), C.c_double, P(Polytope)]),
        "o_step_bodies": (None, [C.c_void_p, P(C.c_uint32), C.c_uint32, P(C.c_double), P(C.c_uint32), C.c_double,
                                 C.c_uint32, P(C.c_uint32), C.c_int]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype, f.argtypes = res, args
    _lib = L
    return L


def polytope(kind, scale=1.0):
    L = load()
    p = Polytope()
    {"cube": L.o_polytope_cube, "tetrahedron": L.o_polytope_tetrahedron,
     "icosahedron": L.o_polytope_icosahedron}[kind](C.byref(p))
    if scale != 1.0:
        L.o_polytope_scale(scale, C.byref(p))
    return p


def step_bodies(bodies, shape_id, verts_xyz, vert_offsets, dt, substeps, want_masks=False, threads=1):
    """oracle batch: for each body solver::step(...).  Returns (new bodies, masks or None)."""
    L = load()
    b = np.array(bodies, dtype=np.float64).reshape(-1, 38).copy()
    n = b.shape[0]
    sid = np.ascontiguousarray(shape_id if shape_id is not None else np.zeros(n), dtype=np.uint32)
    v = np.ascontiguousarray(verts_xyz, dtype=np.float64).reshape(-1)
    if v.size == 0:
        v = np.zeros(3)
    off = np.ascontiguousarray(vert_offsets, dtype=np.uint32)
    masks = np.zeros((substeps, n), dtype=np.uint32) if want_masks else None
    u32p, f64p = C.POINTER(C.c_uint32), C.POINTER(C.c_double)
    L.o_step_bodies(b.ctypes.data, sid.ctypes.data_as(u32p), n, v.ctypes.data_as(f64p), off.ctypes.data_as(u32p),
                    dt, substeps, masks.ctypes.data_as(u32p) if want_masks else None, threads)
    return b, masks


def masks_to_contacts(mask_row):
    """(body, vertex) list in reference push order from one substep's masks."""
    out = []
    for body in np.nonzero(mask_row)[0]:
        m = int(mask_row[body])
        v = 0
        while m:
            if m & 1:
                out.append((int(body), v))
            m >>= 1
            v += 1
    return np.array(out, dtype=np.uint32).reshape(-1, 2)


# ---- body-body contact extension (oracle/xpbd_pairs_oracle.c; parity unpinned) -----------------
class Manifold(C.Structure):
    _fields_ = [("separated", C.c_int32), ("feature", C.c_int32), ("index_a", C.c_uint32), ("index_b", C.c_uint32),
                ("separation", C.c_double), ("query", C.c_double * 3), ("n_points", C.c_uint32),
                ("p_ref", Vec3 * 8), ("p_inc", Vec3 * 8)]

    def points(self):
        return (np.array([self.p_ref[i].np() for i in range(self.n_points)]).reshape(-1, 3),
                np.array([self.p_inc[i].np() for i in range(self.n_points)]).reshape(-1, 3))


FEATURE_FACE_A, FEATURE_FACE_B, FEATURE_EDGES = 0, 1, 2


def sat(fa, fb, pa, pb):
    """op_sat for two (position[3], rotation[4]) frames and two Polytope objects."""
    L = load()
    if not hasattr(L, "_sat_ready"):
        L.op_sat.restype, L.op_sat.argtypes = None, [Frame, Frame, C.POINTER(Polytope), C.POINTER(Polytope), C.POINTER(Manifold)]
        L._sat_ready = True
    m = Manifold()
    L.op_sat(frame(*fa), frame(*fb), C.byref(pa), C.byref(pb), C.byref(m))
    return m


class ContactStats(C.Structure):
    _fields_ = [("n_pairs", C.c_uint64), ("n_touching", C.c_uint64), ("n_points", C.c_uint64)]


def polytopes_array(names_scales):
    """C array of oracle Polytope structs, e.g. [("cube", 1.0), ("tetrahedron", 0.5)]."""
    arr = (Polytope * len(names_scales))()
    for k, (name, scale) in enumerate(names_scales):
        arr[k] = polytope(name, scale)
    return arr


def _contacts_api():
    L = load()
    if not hasattr(L, "_contacts_ready"):
        P = C.POINTER
        L.op_contacts_step.restype = None
        L.op_contacts_step.argtypes = [C.c_void_p, P(C.c_uint32), C.c_uint32, P(Polytope), C.c_double, C.c_uint32,
                                       C.c_double, P(C.c_uint32), P(ContactStats)]
        L.op_broadphase.restype = None
        L.op_broadphase.argtypes = [C.c_void_p, P(C.c_uint32), C.c_uint32, P(Polytope), C.c_double, C.c_double,
                                    P(P(C.c_uint32)), P(P(C.c_uint32))]
        L._contacts_ready = True
    return L


def contacts_step(bodies, shape_id, polys, dt, substeps, pad, want_masks=False):
    """op_contacts_step: returns (new bodies, ground masks or None, stats)."""
    L = _contacts_api()
    b = np.array(bodies, dtype=np.float64).reshape(-1, 38).copy()
    n = b.shape[0]
    sid = np.ascontiguousarray(shape_id if shape_id is not None else np.zeros(n), dtype=np.uint32)
    masks = np.zeros((substeps, n), dtype=np.uint32) if want_masks else None
    st = ContactStats()
    u32p = C.POINTER(C.c_uint32)
    L.op_contacts_step(b.ctypes.data, sid.ctypes.data_as(u32p), n, polys, dt, substeps, pad,
                       masks.ctypes.data_as(u32p) if want_masks else None, C.byref(st))
    return b, masks, st


def broadphase(bodies, shape_id, polys, dt, pad):
    """op_broadphase: (offsets[n+1], neighbours) as numpy arrays."""
    L = _contacts_api()
    b = np.ascontiguousarray(bodies, dtype=np.float64).reshape(-1, 38)
    n = b.shape[0]
    sid = np.ascontiguousarray(shape_id if shape_id is not None else np.zeros(n), dtype=np.uint32)
    off, nb = C.POINTER(C.c_uint32)(), C.POINTER(C.c_uint32)()
    L.op_broadphase(b.ctypes.data, sid.ctypes.data_as(C.POINTER(C.c_uint32)), n, polys, dt, pad, C.byref(off), C.byref(nb))
    offsets = np.ctypeslib.as_array(off, shape=(n + 1,)).copy()
    neigh = np.ctypeslib.as_array(nb, shape=(max(int(offsets[-1]), 1),)).copy()[: offsets[-1]]
    libc = C.CDLL(None)
    libc.free.argtypes = [C.c_void_p]
    libc.free(off)
    libc.free(nb)
    return offsets, neigh


class Joint(C.Structure):
    _fields_ = [("body_a", C.c_uint32), ("body_b", C.c_uint32), ("anchor_a", C.c_double * 3), ("anchor_b", C.c_double * 3),
                ("distance", C.c_double), ("axis_a", C.c_double * 3), ("axis_b", C.c_double * 3), ("kind", C.c_uint32),
                ("reserved", C.c_uint32)]


def _frames_api():
    L = _contacts_api()
    if not hasattr(L, "_frames_ready"):
        P = C.POINTER
        L.op_contacts_begin.restype = C.c_void_p
        L.op_contacts_begin.argtypes = [C.c_void_p, P(C.c_uint32), C.c_uint32, P(Polytope), C.c_double, C.c_double]
        L.op_contacts_attach_joints.restype = None
        L.op_contacts_attach_joints.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32]
        L.op_contacts_substep.restype = None
        L.op_contacts_substep.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        L.op_contacts_end.restype = None
        L.op_contacts_end.argtypes = [C.c_void_p]
        L._frames_ready = True
    return L


def contacts_step_joints(bodies, shape_id, polys, joints, dt, substeps, pad, narrowphase=0, stats=None, max_depenetration_speed=0.0):
    """One frame of op_contacts_* with joints (numpy records with the layout of `Joint`); narrowphase 1 = GJK + EPA.
    stats: optional ContactStats that the substeps add their touching pairs and manifold points to."""
    L = _frames_api()
    b = np.array(bodies, dtype=np.float64).reshape(-1, 38).copy()
    n = b.shape[0]
    sid = np.ascontiguousarray(shape_id if shape_id is not None else np.zeros(n), dtype=np.uint32)
    j = np.ascontiguousarray(joints)
    assert j.dtype.itemsize == C.sizeof(Joint)
    f = L.op_contacts_begin(b.ctypes.data, sid.ctypes.data_as(C.POINTER(C.c_uint32)), n, polys, dt, pad)
    L.op_contacts_attach_joints(f, j.ctypes.data if j.size else None, j.size)
    L.op_contacts_set_narrowphase.restype, L.op_contacts_set_narrowphase.argtypes = None, [C.c_void_p, C.c_int]
    L.op_contacts_set_narrowphase(f, narrowphase)
    L.op_contacts_set_max_depenetration_speed.restype, L.op_contacts_set_max_depenetration_speed.argtypes = None, [C.c_void_p, C.c_double]
    L.op_contacts_set_max_depenetration_speed(f, max_depenetration_speed)
    for _ in range(substeps):
        L.op_contacts_substep(f, b.ctypes.data, dt / substeps, None, C.byref(stats) if stats is not None else None)
    L.op_contacts_end(f)
    return b


# ---- GJK + EPA (oracle/xpbd_gjk_oracle.c; extension, parity unpinned) ---------------------------
class GjkResult(C.Structure):
    _fields_ = [("status", C.c_int32), ("gjk_iterations", C.c_uint32), ("epa_iterations", C.c_uint32),
                ("reserved", C.c_uint32), ("depth", C.c_double), ("normal", Vec3), ("point_a", Vec3), ("point_b", Vec3)]


GJK_SEPARATED, GJK_PENETRATING, GJK_DEGENERATE = 0, 1, 2


def gjk_epa(fa, fb, pa, pb):
    L = load()
    if not hasattr(L, "_gjk_ready"):
        L.og_gjk_epa.restype, L.og_gjk_epa.argtypes = None, [Frame, Frame, C.POINTER(Polytope), C.POINTER(Polytope), C.POINTER(GjkResult)]
        L._gjk_ready = True
    r = GjkResult()
    L.og_gjk_epa(frame(*fa), frame(*fb), C.byref(pa), C.byref(pb), C.byref(r))
    return r


def gjk_epa_cached(fa, fb, pa, pb, axis):
    """og_gjk_epa_cached: `axis` (numpy [3], zero = none) is consulted first; returns (result, refreshed axis)."""
    L = load()
    if not hasattr(L, "_gjk_cached_ready"):
        L.og_gjk_epa_cached.restype = None
        L.og_gjk_epa_cached.argtypes = [Frame, Frame, C.POINTER(Polytope), C.POINTER(Polytope), C.POINTER(Vec3), C.POINTER(GjkResult)]
        L.og_direction_separates.restype = C.c_int
        L.og_direction_separates.argtypes = [Frame, Frame, C.POINTER(Polytope), C.POINTER(Polytope), Vec3]
        L._gjk_cached_ready = True
    r, a = GjkResult(), Vec3(float(axis[0]), float(axis[1]), float(axis[2]))
    L.og_gjk_epa_cached(frame(*fa), frame(*fb), C.byref(pa), C.byref(pb), C.byref(a), C.byref(r))
    return r, a.np()
