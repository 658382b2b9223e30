//! xpbd_ffi.rs -- Rust binding of include/xpbd.h for jim-ec/constraint_solver.
//!
//! NOT COMPILED IN THIS REPOSITORY: the build image has no rustc/cargo.  This is the
//! binding a maintainer of the reference adds (as `src/xpbd_ffi.rs`, plus
//! `println!("cargo:rustc-link-lib=dylib=xpbd_hip")` in build.rs) to run the per-substep
//! hot path on an MI355X.  The compiled and tested equivalent in this repository is the C++
//! host mirror `constraint_solver_amd/host/constraint_solver.hpp`, which calls the very same
//! extern "C" entry points.
//!
//! What it replaces in the reference:
//!   solver::step        src/solver.rs:3-17   -> xpbd_step_one      (literal, one body)
//!   World::integrate    src/world.rs:34-43   -> xpbd_world_step    (batched, state stays in HBM)
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

use cgmath::{Matrix3, Quaternion, Vector3};

use crate::{geometry::Polytope, rigid::Rigid};

/// repr(C) mirror of `Rigid` (src/rigid.rs:6-50), `color` dropped: 38 f64.
/// cgmath's own structs are not repr(C)-stable (Quaternion stores v before s in 0.18),
/// so the conversion is spelled out field by field.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct XpbdRigid {
    pub inverse_mass: f64,
    pub inverse_inertia: [f64; 9], // column-major, [3 * col + row]
    pub external_force: [f64; 3],
    pub internal_force: [f64; 3],
    pub external_torque: [f64; 3],
    pub internal_torque: [f64; 3],
    pub velocity: [f64; 3],
    pub angular_velocity: [f64; 3],
    pub center_of_mass: [f64; 3],
    pub position: [f64; 3],
    pub rotation: [f64; 4], // s, x, y, z  (Quaternion::new argument order)
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct XpbdContact {
    pub body: u32,
    pub vertex: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct XpbdConfig {
    pub struct_size: u32,
    pub device: i32,
    pub mode: u32,
    pub flags: u32,
    pub block_size: u32,
    pub reserved: [u32; 3],
}

/// EXTENSION (not in the reference): full polytope topology for body-body contacts.
#[repr(C)]
pub struct XpbdPolytope {
    pub vertices_xyz: *const f64,
    pub edges: *const u32,
    pub face_offsets: *const u32,
    pub face_indices: *const u32,
    pub n_vertices: u32,
    pub n_edges: u32,
    pub n_faces: u32,
    pub reserved: u32,
    pub centroid: [f64; 3],
}

/// EXTENSION: contact manifold of one body pair.
#[repr(C)]
#[derive(Clone, Copy)]
pub struct XpbdManifold {
    pub n_points: u32,
    pub feature: u32,
    pub index_a: u32,
    pub index_b: u32,
    pub separation: f64,
    pub p_ref: [[f64; 3]; 8],
    pub p_inc: [[f64; 3]; 8],
}

/// EXTENSION: distance / ball joint between two bodies (a hinge is two ball joints on its axis).
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct XpbdJoint {
    pub body_a: u32,
    pub body_b: u32,
    pub anchor_a: [f64; 3],
    pub anchor_b: [f64; 3],
    pub distance: f64,
    pub axis_a: [f64; 3], // XPBD_JOINT_HINGE: unit axes in the object space of a / b that the joint keeps aligned
    pub axis_b: [f64; 3],
    pub kind: u32,        // XPBD_JOINT_DISTANCE = 0, XPBD_JOINT_HINGE = 1
    pub reserved: u32,
}

/// EXTENSION: result of the GJK + EPA narrowphase for one pair.
#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct XpbdGjkResult {
    pub status: i32,
    pub gjk_iterations: u32,
    pub epa_iterations: u32,
    pub reserved: u32,
    pub depth: f64,
    pub normal: [f64; 3],
    pub point_a: [f64; 3],
    pub point_b: [f64; 3],
}

/// xpbd_edge_query: the reference's edge_axes_separation result, (f64::MIN, (usize::MAX, usize::MAX)) as (-DBL_MAX, !0, !0)
#[repr(C)]
pub struct XpbdEdgeQuery {
    pub separation: f64,
    pub edge_a: u32,
    pub edge_b: u32,
}

#[repr(C)]
pub struct XpbdWorld {
    _private: [u8; 0],
}

#[repr(C)]
pub struct XpbdMultiWorld {
    _private: [u8; 0],
}

pub const XPBD_COMM_ID_BYTES: usize = 128;
pub const XPBD_TRANSPORT_RCCL: u32 = 0;
pub const XPBD_TRANSPORT_LOCAL: u32 = 1;
pub const XPBD_MULTI_AUTO_REPLAN: u32 = 1;
pub const XPBD_MULTI_PLAN_THROUGH_DEVICE: u32 = 2;
pub const XPBD_E_HALO: c_int = -7;

/// xpbd_multi_config (include/xpbd.h)
#[repr(C)]
pub struct XpbdMultiConfig {
    pub struct_size: u32,
    pub n_ranks: u32,
    pub first_rank: u32,
    pub n_local: u32,
    pub devices: *const i32,
    pub transport: u32,
    pub flags: u32,
    pub comm_id: *const u8,
    pub contact_pad: f64,
    pub halo_margin: f64,
    pub narrowphase: u32,
    pub reserved: u32,
}

#[link(name = "xpbd_hip")]
extern "C" {
    pub fn xpbd_abi_version() -> u32;
    pub fn xpbd_last_error() -> *const c_char;
    pub fn xpbd_config_default(cfg: *mut XpbdConfig);
    pub fn xpbd_device_count() -> c_int;
    pub fn xpbd_world_create(out: *mut *mut XpbdWorld, cfg: *const XpbdConfig) -> c_int;
    pub fn xpbd_world_destroy(w: *mut XpbdWorld);
    pub fn xpbd_world_set_shapes(w: *mut XpbdWorld, verts_xyz: *const f64, vert_offsets: *const u32, n_shapes: u32) -> c_int;
    pub fn xpbd_world_upload_bodies(w: *mut XpbdWorld, aos: *const XpbdRigid, shape_id: *const u32, n: u32) -> c_int;
    pub fn xpbd_world_download_bodies(w: *mut XpbdWorld, aos: *mut XpbdRigid, n: u32) -> c_int;
    pub fn xpbd_world_body_count(w: *const XpbdWorld) -> u32;
    pub fn xpbd_world_step(w: *mut XpbdWorld, dt: f64, substeps: u32) -> c_int;
    pub fn xpbd_world_synchronize(w: *mut XpbdWorld) -> c_int;
    pub fn xpbd_world_download_contacts(w: *mut XpbdWorld, out: *mut XpbdContact, cap: u32, n_out: *mut u32) -> c_int;
    pub fn xpbd_world_download_contact_masks(w: *mut XpbdWorld, masks: *mut u32, substeps: u32, n: u32) -> c_int;
    pub fn xpbd_world_set_stream(w: *mut XpbdWorld, hip_stream: *mut c_void) -> c_int;
    pub fn xpbd_world_get_stream(w: *const XpbdWorld) -> *mut c_void;
    pub fn xpbd_world_set_mode(w: *mut XpbdWorld, mode: u32) -> c_int;
    pub fn xpbd_step_one(rigid: *mut XpbdRigid, verts_xyz: *const f64, nverts: u32, dt: f64, substeps: u32) -> c_int;
    pub fn xpbd_selftest_div_sqrt(device: i32, a: *const f64, b: *const f64, q: *mut f64, r: *mut f64, n: u32) -> c_int;
    pub fn xpbd_selftest_hbm_copy(device: i32, bytes: u64, repeats: u32, gbytes_per_s: *mut f64) -> c_int;
    pub fn xpbd_selftest_field_streams(device: i32, bodies: u64, tile_major: u32, repeats: u32, gbytes_per_s: *mut f64) -> c_int;
    pub fn xpbd_selftest_gather(device: i32, records: u32, record_bytes: u32, read_bytes: u32, repeats: u32, gbytes_per_s: *mut f64) -> c_int;
    pub fn xpbd_world_download_frames(w: *mut XpbdWorld, frames: *mut f64, n: u32) -> c_int;
    // ---- extension: body-body contacts, joints, multi-GPU halo exchange (not in the reference) ----
    pub fn xpbd_world_set_polytopes(w: *mut XpbdWorld, shapes: *const XpbdPolytope, n_shapes: u32) -> c_int;
    pub fn xpbd_world_narrowphase(w: *mut XpbdWorld, pairs: *const u32, n_pairs: u32, out: *mut XpbdManifold) -> c_int;
    pub fn xpbd_world_edge_axes_separation(w: *mut XpbdWorld, pairs: *const u32, n_pairs: u32, out: *mut XpbdEdgeQuery) -> c_int;
    pub fn xpbd_world_narrowphase_gjk(w: *mut XpbdWorld, pairs: *const u32, n_pairs: u32, out: *mut XpbdGjkResult) -> c_int;
    pub fn xpbd_world_set_narrowphase(w: *mut XpbdWorld, narrowphase: u32) -> c_int;
    pub fn xpbd_world_set_sat_schedule(w: *mut XpbdWorld, schedule: u32) -> c_int;
    pub fn xpbd_world_set_contact_pad(w: *mut XpbdWorld, pad: f64) -> c_int;
    pub fn xpbd_world_contact_stats(w: *mut XpbdWorld, out: *mut u64) -> c_int;
    pub fn xpbd_world_build_neighbours(w: *mut XpbdWorld, dt: f64, n_entries_out: *mut u32) -> c_int;
    pub fn xpbd_world_download_neighbours(w: *mut XpbdWorld, offsets: *mut u32, neighbours: *mut u32, cap: u32) -> c_int;
    pub fn xpbd_world_set_joints(w: *mut XpbdWorld, joints: *const XpbdJoint, n_joints: u32) -> c_int;
    pub fn xpbd_world_set_max_depenetration_speed(w: *mut XpbdWorld, speed: f64) -> c_int;
    pub fn xpbd_multi_world_set_max_depenetration_speed(mw: *mut XpbdMultiWorld, speed: f64) -> c_int;
    pub fn xpbd_world_snapshot_positions(w: *mut XpbdWorld, dev_indices: *const u32, n: u32, dev_snapshot: *mut f64) -> c_int;
    pub fn xpbd_world_max_displacement2(w: *mut XpbdWorld, dev_indices: *const u32, n: u32, dev_snapshot: *const f64, dev_scale: *const f64, dev_max: *mut f64) -> c_int;
    // ---- extension: the multi-GPU world (one call per frame; the library owns streams, RCCL communicators and the halo plan) ----
    pub fn xpbd_comm_unique_id(id: *mut u8) -> c_int;
    pub fn xpbd_comm_library() -> *const c_char;
    pub fn xpbd_multi_config_default(cfg: *mut XpbdMultiConfig);
    pub fn xpbd_multi_world_create(out: *mut *mut XpbdMultiWorld, cfg: *const XpbdMultiConfig) -> c_int;
    pub fn xpbd_multi_world_destroy(mw: *mut XpbdMultiWorld);
    pub fn xpbd_multi_world_set_polytopes(mw: *mut XpbdMultiWorld, shapes: *const XpbdPolytope, n_shapes: u32) -> c_int;
    pub fn xpbd_multi_world_upload(mw: *mut XpbdMultiWorld, bodies: *const XpbdRigid, shape_id: *const u32, first_global: u32, n_bodies: u32,
                                   n_global: u32, joints: *const XpbdJoint, n_joints: u32) -> c_int;
    pub fn xpbd_multi_world_step(mw: *mut XpbdMultiWorld, dt: f64, substeps: u32) -> c_int;
    pub fn xpbd_multi_world_replan(mw: *mut XpbdMultiWorld) -> c_int;
    pub fn xpbd_multi_world_synchronize(mw: *mut XpbdMultiWorld) -> c_int;
    pub fn xpbd_multi_world_download(mw: *mut XpbdMultiWorld, out: *mut XpbdRigid, n: u32) -> c_int;
    pub fn xpbd_multi_world_download_owned(mw: *mut XpbdMultiWorld, ids: *mut u32, out: *mut XpbdRigid, cap: u32, n_out: *mut u32) -> c_int;
    pub fn xpbd_multi_world_halo_stats(mw: *mut XpbdMultiWorld, out: *mut u64, max_displacement: *mut f64) -> c_int;
    pub fn xpbd_multi_world_plan_stats(mw: *mut XpbdMultiWorld, out: *mut u64) -> c_int; // out: [u64; 12]
    pub fn xpbd_multi_world_owners(mw: *mut XpbdMultiWorld, owner: *mut u8, n_global: u32) -> c_int;
    pub fn xpbd_multi_world_contact_stats(mw: *mut XpbdMultiWorld, out: *mut u64) -> c_int;
    pub fn xpbd_halo_cell_key(centre: *const f64, cell_edge: f64) -> i64;
    pub fn xpbd_halo_partition(cell_keys: *const i64, n_global: u32, n_ranks: u32, owner: *mut u8) -> c_int;
    pub fn xpbd_halo_plan_owned(cell_keys: *const i64, owner: *const u8, n_global: u32, n_ranks: u32, rank: u32, joints: *const XpbdJoint,
                                n_joints: u32, ghosts: *mut u32, n_ghosts: *mut u32, boundary: *mut u32, n_boundary: *mut u32, far: *mut u8,
                                cap: u32) -> c_int;
    pub fn xpbd_halo_plan_light(keys_at_cut: *const i64, cell_keys: *const i64, n_global: u32, n_ranks: u32, rank: u32, joints: *const XpbdJoint,
                                n_joints: u32, owner_now: *mut u8, own: *mut u32, n_own: *mut u32, ghosts: *mut u32, n_ghosts: *mut u32,
                                boundary: *mut u32, n_boundary: *mut u32, far: *mut u8, cap: u32) -> c_int;
    pub fn xpbd_halo_plan_far(cell_keys: *const i64, n_global: u32, n_ranks: u32, rank: u32, far: *mut u8, cap: u32, n_owned: *mut u32) -> c_int;
    pub fn xpbd_halo_plan(cell_keys: *const i64, n_global: u32, n_ranks: u32, rank: u32, joints: *const XpbdJoint, n_joints: u32,
                          ghosts: *mut u32, n_ghosts: *mut u32, boundary: *mut u32, n_boundary: *mut u32, cap: u32) -> c_int;
    pub fn xpbd_world_contacts_begin(w: *mut XpbdWorld, dt: f64) -> c_int;
    pub fn xpbd_world_contacts_substep(w: *mut XpbdWorld, h: f64) -> c_int;
    pub fn xpbd_world_export_dynamic(w: *mut XpbdWorld, dev_indices: *const u32, n: u32, dev_buf: *mut f64) -> c_int;
    pub fn xpbd_world_import_dynamic(w: *mut XpbdWorld, dev_indices: *const u32, n: u32, dev_buf: *const f64) -> c_int;
    pub fn xpbd_world_import_dynamic_rows(w: *mut XpbdWorld, dev_indices: *const u32, dev_rows: *const u32, n: u32, dev_buf: *const f64) -> c_int;
    // state history: replaces `states: Vec<(World, DebugLines)>` of src/app.rs:48 (see HistoryWorld below)
    pub fn xpbd_world_history_push(w: *mut XpbdWorld, index_out: *mut u32) -> c_int;
    pub fn xpbd_world_history_restore(w: *mut XpbdWorld, index: u32) -> c_int;
    pub fn xpbd_world_history_truncate(w: *mut XpbdWorld, length: u32) -> c_int;
    pub fn xpbd_world_history_length(w: *const XpbdWorld) -> u32;
}

fn v3(v: Vector3<f64>) -> [f64; 3] {
    [v.x, v.y, v.z]
}

fn from3(a: [f64; 3]) -> Vector3<f64> {
    Vector3::new(a[0], a[1], a[2])
}

impl From<&Rigid> for XpbdRigid {
    fn from(r: &Rigid) -> Self {
        let m: &Matrix3<f64> = &r.inverse_inertia;
        XpbdRigid {
            inverse_mass: r.inverse_mass,
            inverse_inertia: [m.x.x, m.x.y, m.x.z, m.y.x, m.y.y, m.y.z, m.z.x, m.z.y, m.z.z],
            external_force: v3(r.external_force),
            internal_force: v3(r.internal_force),
            external_torque: v3(r.external_torque),
            internal_torque: v3(r.internal_torque),
            velocity: v3(r.velocity),
            angular_velocity: v3(r.angular_velocity),
            center_of_mass: v3(r.center_of_mass),
            position: v3(r.position),
            rotation: [r.rotation.s, r.rotation.v.x, r.rotation.v.y, r.rotation.v.z],
        }
    }
}

impl XpbdRigid {
    /// Writes the dynamic state back; the static fields were never changed by the device.
    pub fn store_into(&self, r: &mut Rigid) {
        r.velocity = from3(self.velocity);
        r.angular_velocity = from3(self.angular_velocity);
        r.position = from3(self.position);
        r.rotation = Quaternion::new(self.rotation[0], self.rotation[1], self.rotation[2], self.rotation[3]);
    }
}

fn check(rc: c_int) {
    if rc != 0 {
        let msg = unsafe { CStr::from_ptr(xpbd_last_error()) }.to_string_lossy().into_owned();
        // the reference panics on its own failure paths (src/rigid.rs:59); keep that contract
        panic!("xpbd error {}: {}", rc, msg);
    }
}

fn flat_vertices(polytope: &Polytope) -> Vec<f64> {
    polytope.vertices.iter().flat_map(|v| [v.x, v.y, v.z]).collect()
}

/// Drop-in body for `solver::step` (src/solver.rs:3): same signature, same result.
pub fn step(rigid: &mut Rigid, polytope: &Polytope, dt: f64, substep_count: usize) {
    let mut c = XpbdRigid::from(&*rigid);
    let verts = flat_vertices(polytope);
    check(unsafe { xpbd_step_one(&mut c, verts.as_ptr(), polytope.vertices.len() as u32, dt, substep_count as u32) });
    c.store_into(rigid);
}

/// N-body world resident on one GPU: upload once, `integrate` every frame, read poses back for rendering.
pub struct GpuWorld {
    handle: *mut XpbdWorld,
    staging: Vec<XpbdRigid>,
}

impl GpuWorld {
    pub fn new(device: i32, shapes: &[&Polytope], bodies: &[Rigid], shape_id: &[u32]) -> GpuWorld {
        let mut cfg = XpbdConfig::default();
        unsafe { xpbd_config_default(&mut cfg) };
        cfg.device = device;
        let mut handle = std::ptr::null_mut();
        check(unsafe { xpbd_world_create(&mut handle, &cfg) });
        let mut verts = Vec::new();
        let mut offsets = vec![0u32];
        for p in shapes {
            verts.extend(flat_vertices(p));
            offsets.push((verts.len() / 3) as u32);
        }
        check(unsafe { xpbd_world_set_shapes(handle, verts.as_ptr(), offsets.as_ptr(), shapes.len() as u32) });
        let staging: Vec<XpbdRigid> = bodies.iter().map(XpbdRigid::from).collect();
        check(unsafe { xpbd_world_upload_bodies(handle, staging.as_ptr(), shape_id.as_ptr(), staging.len() as u32) });
        GpuWorld { handle, staging }
    }

    /// World::integrate (src/world.rs:34-43) for every body: solver::step(body, shape, dt, substeps).
    pub fn integrate(&mut self, dt: f64, substeps: u32) {
        check(unsafe { xpbd_world_step(self.handle, dt, substeps) });
    }

    /// Read-back for rendering (src/app.rs:227-230 consumes Rigid::frame()).
    pub fn read_back(&mut self, bodies: &mut [Rigid]) {
        check(unsafe { xpbd_world_download_bodies(self.handle, self.staging.as_mut_ptr(), self.staging.len() as u32) });
        for (c, r) in self.staging.iter().zip(bodies.iter_mut()) {
            c.store_into(r);
        }
    }
}

/// The app's state history (src/app.rs:48, 206-212) on the device: `states` becomes the world's history,
/// `current_state` stays a host-side cursor.  `advance` is the body of the `RedrawRequested` loop.
pub struct Timeline {
    pub world: GpuWorld,
    pub current_state: u32,
}

impl Timeline {
    /// `states = vec![(World::new(..), ..)]; current_state = 0`
    pub fn new(world: GpuWorld) -> Timeline {
        check(unsafe { xpbd_world_history_push(world.handle, std::ptr::null_mut()) });
        Timeline { world, current_state: 0 }
    }

    /// `for _ in 0..time_speed() { if current_state + 1 >= states.len() { integrate; push } current_state += 1 }`
    pub fn advance(&mut self, dt: f64, substeps: u32, time_speed: u32) {
        for _ in 0..time_speed {
            let len = unsafe { xpbd_world_history_length(self.world.handle) };
            if self.current_state + 1 >= len {
                check(unsafe { xpbd_world_history_restore(self.world.handle, len - 1) });
                self.world.integrate(dt, substeps);
                check(unsafe { xpbd_world_history_push(self.world.handle, std::ptr::null_mut()) });
            }
            self.current_state += 1;
        }
    }

    /// Scrubbing: show state `index` (the renderer then calls `world.read_back`).
    pub fn seek(&mut self, index: u32) {
        check(unsafe { xpbd_world_history_restore(self.world.handle, index) });
        self.current_state = index;
    }
}

impl Drop for GpuWorld {
    fn drop(&mut self) {
        unsafe { xpbd_world_destroy(self.handle) };
    }
}
