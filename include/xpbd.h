/*
 * xpbd.h -- C ABI of the MI355X-native XPBD rigid-body stepper.
 *
 * This is the drop-in boundary for the per-substep hot path of
 * jim-ec/constraint_solver: everything below `solver::step`
 * (reference src/solver.rs:3-17) runs in hand-written HIP kernels for gfx950;
 * everything above it (World, Rigid, Polytope construction, the app) stays on
 * the host and calls these entry points.  Plain pointers and sizes only.
 *
 * Reference interface each entry point replaces:
 *   xpbd_step_one               solver::step(&mut Rigid,&Polytope,dt,n)   src/solver.rs:3
 *   xpbd_world_step             World::integrate's loop of solver::step    src/world.rs:34-43
 *   xpbd_world_upload_bodies    the `&mut Rigid` borrows of that loop      src/world.rs:41-42, src/rigid.rs:6-50
 *   xpbd_world_set_shapes       the `&Polytope` borrow (vertices only)     src/solver.rs:3, src/geometry.rs:82-93
 *   xpbd_world_download_bodies  Rigid read-back (frame() for rendering)    src/app.rs:227-230, src/rigid.rs:75-80
 *   xpbd_world_download_contacts  the Vec<Constraint> push order of ground src/collision.rs:16-32
 *
 * Semantics: for every body i,  xpbd_world_step(w, dt, n)  ==
 *   solver::step(&mut body[i], &shape[shape_id[i]], dt, n)
 * in IEEE f64 with the reference's operation order (no FMA contraction), so
 * ground-contact index lists are bit-exact and poses agree with the CPU
 * reference (target 1e-5 relative; observed bit-identical).
 *
 * Threading: one xpbd_world is used from one thread at a time; distinct worlds
 * are independent.  step() enqueues on the world's HIP stream and returns;
 * download_* and synchronize() wait (XPBD_MODE_CONTACTS: step() waits ONCE, for the
 * broadphase's pair counts, which size the pair buffers; everything after that is
 * enqueued).  No call throws or unwinds across the ABI.
 */
#ifndef XPBD_H
#define XPBD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: xpbd_multi_world_* (library-owned sharding), XPBD_E_HALO, edge_axes_separation, state history */
#define XPBD_ABI_VERSION 2u

/* Error codes (reference has no Result on this path; it panics). */
#define XPBD_OK                   0
#define XPBD_E_INVALID          (-1)  /* bad argument / state */
#define XPBD_E_HIP              (-2)  /* a HIP runtime call failed */
#define XPBD_E_OOM              (-3)  /* host or device allocation failed */
#define XPBD_E_SINGULAR_INERTIA (-4)  /* mirrors the panic at src/rigid.rs:59 */
#define XPBD_E_NO_DEVICE        (-5)  /* no usable gfx950 device */
#define XPBD_E_CAPACITY         (-6)  /* caller buffer too small */
#define XPBD_E_HALO              (-7)  /* multi-GPU: a body outran halo_margin since the halos were planned */

/* A shape may have at most this many vertices (contact set is a u32 mask). */
#define XPBD_MAX_SHAPE_VERTS 32u

/*
 * repr(C) mirror of the reference's `Rigid` (src/rigid.rs:6-50), field order
 * kept, `color` dropped: 38 doubles = 304 bytes.
 *   inverse_inertia : cgmath Matrix3, column-major, [3*col + row]
 *   rotation        : cgmath Quaternion::new(w, xi, yj, zk) order: {s, x, y, z}
 */
typedef struct xpbd_rigid {
    double inverse_mass;
    double inverse_inertia[9];
    double external_force[3];
    double internal_force[3];
    double external_torque[3];
    double internal_torque[3];
    double velocity[3];
    double angular_velocity[3];
    double center_of_mass[3];
    double position[3];
    double rotation[4];
} xpbd_rigid;

/* One ground constraint of the last substep: body index and the index of the
 * shape vertex that produced it (src/collision.rs:16-32 push order). */
typedef struct xpbd_contact {
    uint32_t body;
    uint32_t vertex;
} xpbd_contact;

/* How xpbd_world_step schedules the substep loop. */
#define XPBD_MODE_FUSED        0u /* one launch runs all substeps in registers (bodies are independent) */
#define XPBD_MODE_PER_SUBSTEP  1u /* one launch per substep: state round-trips HBM each substep */
#define XPBD_MODE_CONTACTS     2u /* EXTENSION: ground + body-body contacts (needs xpbd_world_set_polytopes) */

#define XPBD_FLAG_TRACE_CONTACTS 1u /* keep the contact mask of every substep of the last step() call */

typedef struct xpbd_config {
    uint32_t struct_size;  /* = sizeof(xpbd_config) */
    int32_t  device;       /* HIP device ordinal */
    uint32_t mode;         /* XPBD_MODE_* */
    uint32_t flags;        /* XPBD_FLAG_* */
    uint32_t block_size;   /* threads per workgroup, multiple of 64, <= 256; 0 = library default (64) */
    uint32_t reserved[3];  /* must be 0 */
} xpbd_config;

typedef struct xpbd_world xpbd_world;

/* Library / error ---------------------------------------------------------- */
uint32_t    xpbd_abi_version(void);
/* Message for the last failing call on this thread; valid until the next call. */
const char *xpbd_last_error(void);
/* Fills cfg with defaults (device 0, fused mode, no flags). */
void        xpbd_config_default(xpbd_config *cfg);
/* Number of visible HIP devices, or a negative error code. */
int         xpbd_device_count(void);

/* World lifetime ----------------------------------------------------------- */
int  xpbd_world_create(xpbd_world **out, const xpbd_config *cfg);
void xpbd_world_destroy(xpbd_world *w);

/* Shapes: all shapes' vertices back to back (xyz triples) and a CSR offset
 * array of n_shapes+1 entries (in vertices).  Copied; caller keeps ownership.
 * While bodies are resident the new table must still cover every shape id they use (else XPBD_E_INVALID). */
int  xpbd_world_set_shapes(xpbd_world *w, const double *verts_xyz,
                           const uint32_t *vert_offsets, uint32_t n_shapes);

/* Bodies: AoS host array -> SoA device layout.  shape_id may be NULL (all 0). */
int  xpbd_world_upload_bodies(xpbd_world *w, const xpbd_rigid *aos,
                              const uint32_t *shape_id, uint32_t n);
int  xpbd_world_download_bodies(xpbd_world *w, xpbd_rigid *aos, uint32_t n);
uint32_t xpbd_world_body_count(const xpbd_world *w);
/* Rigid::frame() of every body (src/rigid.rs:75-80), the only thing the reference's renderer reads per
 * frame (src/app.rs:227-230): frames[7*i .. 7*i+6] = origin x y z, rotation s x y z.  56 B/body instead of 304. */
int  xpbd_world_download_frames(xpbd_world *w, double *frames, uint32_t n);

/* for each body: solver::step(body, shape[body], dt, substeps).  Asynchronous. */
int  xpbd_world_step(xpbd_world *w, double dt, uint32_t substeps);
int  xpbd_world_synchronize(xpbd_world *w);

/* Contacts of the LAST substep of the last step(), sorted by body then vertex
 * (= reference push order).  *n_out receives the total count even when it
 * exceeds cap (then XPBD_E_CAPACITY is returned and out holds the first cap). */
int  xpbd_world_download_contacts(xpbd_world *w, xpbd_contact *out, uint32_t cap,
                                  uint32_t *n_out);
/* With XPBD_FLAG_TRACE_CONTACTS: masks[k*n + i] = bit set of shape vertices of
 * body i that produced a constraint in substep k of the last step() call. */
int  xpbd_world_download_contact_masks(xpbd_world *w, uint32_t *masks,
                                       uint32_t substeps, uint32_t n);

/* Stream interop: run on a caller-owned hipStream_t (NULL restores the
 * world's own stream).  The caller keeps the stream alive. */
int  xpbd_world_set_stream(xpbd_world *w, void *hip_stream);
void *xpbd_world_get_stream(const xpbd_world *w);
int  xpbd_world_set_mode(xpbd_world *w, uint32_t mode);

/* Literal single-body drop-in for solver::step (src/solver.rs:3): uploads,
 * steps on device 0 and downloads one body.  verts: nverts xyz triples. */
int  xpbd_step_one(xpbd_rigid *rigid, const double *verts_xyz, uint32_t nverts,
                   double dt, uint32_t substeps);

/* ---------------------------------------------------------------------------
 * EXTENSION -- body-body contacts.  NOT in the reference:
 * its `sat` (src/collision.rs:37-121) is an uncalled stub and `World` holds two
 * bodies that never interact.  These entry points finish that sketch; parity
 * for them is unpinned (own CPU oracle + invariants), and nothing above
 * changes behaviour when they are not used.
 * ------------------------------------------------------------------------- */

/* Full convex polytope (reference `Polytope`, src/geometry.rs:82-93): vertices,
 * edges (vertex index pairs), faces as CSR (face_offsets has n_faces+1 entries
 * into face_indices) and the centroid.  Faces need 3..8 vertices. */
typedef struct xpbd_polytope {
    const double   *vertices_xyz;
    const uint32_t *edges;
    const uint32_t *face_offsets;
    const uint32_t *face_indices;
    uint32_t n_vertices, n_edges, n_faces, reserved;
    double   centroid[3];
} xpbd_polytope;

#define XPBD_MAX_MANIFOLD_POINTS 8u
#define XPBD_FEATURE_FACE_A 0u  /* reference face on A, incident body B */
#define XPBD_FEATURE_FACE_B 1u  /* reference face on B, incident body A */
#define XPBD_FEATURE_EDGES  2u  /* edge of A (reference) against edge of B */

/* Contact manifold of one pair.  p_ref[k] lies on the reference body's surface,
 * p_inc[k] is the penetrating point of the incident body (world space). */
typedef struct xpbd_manifold {
    uint32_t n_points;   /* 0: separated */
    uint32_t feature;    /* XPBD_FEATURE_* (valid when n_points > 0) */
    uint32_t index_a;    /* face of A (reference or incident) or edge of A */
    uint32_t index_b;    /* face of B (incident or reference) or edge of B */
    double   separation; /* largest separating-axis value, < 0 */
    double   p_ref[XPBD_MAX_MANIFOLD_POINTS][3];
    double   p_inc[XPBD_MAX_MANIFOLD_POINTS][3];
} xpbd_manifold;

/* Replaces xpbd_world_set_shapes when body-body contacts are wanted: sets the
 * vertex tables AND the face/edge topology (outward face planes are derived as
 * Polytope::plane does, src/geometry.rs:262-271). */
int  xpbd_world_set_polytopes(xpbd_world *w, const xpbd_polytope *shapes, uint32_t n_shapes);

/* SAT narrowphase of the given body pairs (pairs[2k], pairs[2k+1] = A, B) at the
 * world's current poses: a group of 16, 32 or 64 lanes per pair (by the largest shape).
 * out has n_pairs entries. */
int  xpbd_world_narrowphase(xpbd_world *w, const uint32_t *pairs, uint32_t n_pairs,
                            xpbd_manifold *out);

/* The reference's edge_axes_separation (src/collision.rs:151-197: written, but called by nobody there, not even by `sat`)
 * for the given pairs at the world's current poses, literally: all E_A x E_B edge pairs, axis = normalize(eA x eB) turned
 * away from A's centroid, skipped when A has a vertex beyond the edge's foot, distance of B's support along -axis; first
 * maximum wins, parallel edges (NaN axis) contribute nothing.  Returns (f64::MIN, (usize::MAX, usize::MAX)) as
 * (-DBL_MAX, 0xFFFFFFFF, 0xFFFFFFFF).  Diagnostic: the contact pipeline's SAT tests the unique edge DIRECTIONS instead. */
typedef struct xpbd_edge_query {
    double   separation;
    uint32_t edge_a, edge_b;
} xpbd_edge_query;
int  xpbd_world_edge_axes_separation(xpbd_world *w, const uint32_t *pairs, uint32_t n_pairs, xpbd_edge_query *out);

/* GJK + EPA narrowphase of the given pairs (the reference has neither): boolean GJK on the
 * Minkowski difference built with the reference's support convention (src/geometry.rs:274-289), then EPA for the
 * penetration depth, the normal (from A to B) and one witness point on each body.  16 or 32 lanes per pair
 * for the boolean GJK, one wave per penetrating pair for EPA. */
#define XPBD_GJK_SEPARATED   0
#define XPBD_GJK_PENETRATING 1
#define XPBD_GJK_DEGENERATE  2  /* origin on the simplex boundary / flat simplex / iteration cap: use the SAT */
typedef struct xpbd_gjk_result {
    int32_t  status;
    uint32_t gjk_iterations, epa_iterations, reserved;
    double   depth;
    double   normal[3];
    double   point_a[3];
    double   point_b[3];   /* point_a - point_b = depth * normal */
} xpbd_gjk_result;
int  xpbd_world_narrowphase_gjk(xpbd_world *w, const uint32_t *pairs, uint32_t n_pairs, xpbd_gjk_result *out);
/* Narrowphase used by XPBD_MODE_CONTACTS: the SAT (default: up to 8 points per pair) or GJK + EPA (a clipped face
 * contact where the penetration normal is a face normal of one of the bodies, else one point with reference body A /
 * incident body B; a degenerate query gives no contact in that substep).  In the pipeline a pair that GJK finds
 * separated keeps the direction that proved it, and the pair's next query within the same step call tries that
 * support plane first (a settled pile: most separated pairs stay separated by the same plane). */
#define XPBD_NARROWPHASE_SAT     0u
#define XPBD_NARROWPHASE_GJK_EPA 1u
int  xpbd_world_set_narrowphase(xpbd_world *w, uint32_t narrowphase);
/* In XPBD_MODE_CONTACTS both narrowphases first test the TIGHT bounding spheres of a pair (centroid, largest
 * vertex distance): the neighbour lists are built once per step call from spheres inflated by a whole frame of
 * travel, so in a given substep most pairs of a loose scene cannot touch, and those are answered "no contact"
 * without running the query.  (xpbd_world_narrowphase / _gjk, the diagnostic entry points, always run it.)
 * For the SAT the test can run inside the narrowphase kernel (ONE_PASS) or as a pass of its own followed by the SAT
 * over the surviving pairs only (TWO_PASS; that pass also answers the pairs still separated by the face axis that
 * separated them last time -- the SAT's own arithmetic for that face): identical results, different cost -- AUTO
 * picks per step call from the share of touching pairs in the previous one.  GJK + EPA always runs it as a pass. */
#define XPBD_SAT_SCHEDULE_AUTO     0u
#define XPBD_SAT_SCHEDULE_ONE_PASS 1u
#define XPBD_SAT_SCHEDULE_TWO_PASS 2u
int  xpbd_world_set_sat_schedule(xpbd_world *w, uint32_t schedule);

/* XPBD_MODE_CONTACTS: per xpbd_world_step a sphere broadphase builds sorted neighbour lists
 * (sphere = centroid, r_shape + min(|v| dt, r_shape) + pad); per substep: integrate -> SAT of every neighbour
 * pair -> ground contacts (reference path) -> pair contacts, Jacobi-averaged with a fixed
 * summation order -> derive.  Exact semantics: oracle/xpbd_pairs_oracle.h.  With no overlapping
 * spheres the result equals XPBD_MODE_PER_SUBSTEP bit for bit. */
int  xpbd_world_set_contact_pad(xpbd_world *w, double pad);            /* default 0.02 (metres) */
/* Optional limit on how fast a BODY-BODY contact may push its bodies apart: the length of a contact point's positional
 * correction is limited to max(0, speed * h - what the incident point has already moved towards the reference surface in
 * this substep) before lambda is formed, so the bodies part at `speed` instead of accelerating (0 = off, the default = the reference's solver loop,
 * src/solver.rs:19-27, which resolves any penetration within ONE substep, i.e. at depth / h -- 120 m/s for 0.1 m at 20
 * substeps per frame; light bodies squeezed between heavy ones leave a pile at that speed).  With the knob on, the ground
 * contacts of XPBD_MODE_CONTACTS are limited the same way; joints never are; XPBD_MODE_FUSED / _PER_SUBSTEP (the reference path)
 * ignore it.  Semantics: oracle/xpbd_pairs_oracle.h. */
int  xpbd_world_set_max_depenetration_speed(xpbd_world *w, double speed);
/* out = {neighbour pairs of the last step, touching pairs, manifold points}; the last two are
 * summed over substeps since the previous call and then reset. */
int  xpbd_world_contact_stats(xpbd_world *w, uint64_t out[3]);
/* Broadphase alone (as the next step(dt, .) would run it) and its CSR neighbour lists. */
int  xpbd_world_build_neighbours(xpbd_world *w, double dt, uint32_t *n_entries_out);
int  xpbd_world_download_neighbours(xpbd_world *w, uint32_t *offsets, uint32_t *neighbours, uint32_t cap);

/* Joints (EXTENSION; the reference has no joint type, only the unused `distance`
 * field of Constraint, src/constraint.rs:9).  A joint keeps |frame_b * anchor_b - frame_a * anchor_a|
 * at `distance` (anchors in object space, the space of the shape vertices).  distance = 0 is a ball
 * joint; XPBD_JOINT_HINGE adds an angular term (below).  Joints are projected in XPBD_MODE_CONTACTS together
 * with the body-body contacts (same Jacobi pass, after a body's contacts, ascending joint index).
 * Body indices refer to the bodies uploaded last; uploading bodies again clears the joints.
 * Only XPBD_MODE_CONTACTS projects joints: in the other modes a non-empty list is XPBD_E_INVALID. */
#define XPBD_JOINT_DISTANCE 0u  /* positional term only (distance = 0: ball joint) */
#define XPBD_JOINT_HINGE    1u  /* positional term + ANGULAR term: the unit axes axis_a / axis_b (object space of a / b) are kept
                                 * aligned.  With a_w = rot_a * axis_a, b_w = rot_b * axis_b: delta = a_w x b_w, n = delta / |delta|,
                                 * w = sum over both bodies of (I^-1 (q^-1 n)) . (q^-1 n) (the angular half of
                                 * Constraint::inverse_resitance, src/constraint.rs:25-32), lambda = |delta| / (w + compliance);
                                 * a turns by +lambda n, b by -lambda n, applied as Rigid::apply_impulse applies an angular
                                 * displacement (src/rigid.rs:118-122).  It is a Jacobi entry of its own, after the joint's
                                 * positional term.  A hinge = ball joint (distance 0) on the axis + this: one degree of freedom. */
typedef struct xpbd_joint {
    uint32_t body_a, body_b;
    double   anchor_a[3];
    double   anchor_b[3];
    double   distance;
    double   axis_a[3];   /* XPBD_JOINT_HINGE: unit vectors */
    double   axis_b[3];
    uint32_t kind;        /* XPBD_JOINT_* */
    uint32_t reserved;    /* must be 0 */
} xpbd_joint;
int  xpbd_world_set_joints(xpbd_world *w, const xpbd_joint *joints, uint32_t n_joints);

/* Split form of xpbd_world_step(w, dt, n) in XPBD_MODE_CONTACTS, for hosts that exchange halo
 * bodies between substeps (multi-GPU):  begin(dt); n x { substep(dt / n); <exchange> }.
 * begin runs the broadphase for the coming frame; substep is one substep of the pipeline. */
int  xpbd_world_contacts_begin(xpbd_world *w, double dt);
int  xpbd_world_contacts_substep(xpbd_world *w, double h);
/* Halo exchange helpers.  dev_indices: DEVICE pointer to n body indices; dev_buf: DEVICE pointer
 * to n x 13 doubles, body-major, in the SoA field order position[3] rotation{s,x,y,z}
 * velocity[3] angular_velocity[3].  Both run asynchronously on the world's stream, so a
 * collective enqueued on the same stream (or ordered after it) sees the data. */
int  xpbd_world_export_dynamic(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, double *dev_buf);
int  xpbd_world_import_dynamic(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, const double *dev_buf);
/* As import_dynamic, but body dev_indices[k] takes row dev_rows[k] of dev_buf: the output of an all-gather (every
 * rank's boundary bodies back to back) is imported as it is, without a gather pass of the host framework. */
int  xpbd_world_import_dynamic_rows(xpbd_world *w, const uint32_t *dev_indices, const uint32_t *dev_rows, uint32_t n,
                                    const double *dev_buf);

/* Halo validity for hosts that run their own exchange loop: dev_snapshot[3k..3k+2] = position of body dev_indices[k] (at
 * plan time), and later *dev_max = max(*dev_max, max_k scale_k * |position_k - snapshot_k|^2) -- the largest squared distance
 * any listed body has travelled (a NaN position counts as +inf); dev_scale is optional (NULL: 1), e.g. 1 / allowance_k^2,
 * which makes the result the largest fraction of its allowance any body has used.  Device pointers; asynchronous on the
 * world's stream. */
int  xpbd_world_snapshot_positions(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, double *dev_snapshot);
int  xpbd_world_max_displacement2(xpbd_world *w, const uint32_t *dev_indices, uint32_t n, const double *dev_snapshot,
                                  const double *dev_scale, double *dev_max);

/* ---------------------------------------------------------------------------
 * Multi-GPU world (EXTENSION): the caller is still World::integrate (src/world.rs:34-43), now over an N-body
 * world in XPBD_MODE_CONTACTS whose bodies are sharded over the GPUs of one node.  The caller numbers its bodies as it likes:
 * OWNERSHIP IS THE LIBRARY'S.  The bodies are binned into a uniform grid (cell edge = 2 * (largest bounding radius +
 * contact_pad + halo_margin)), the cells are ordered by spatial-hash cell key, longest axis of the world first, and that
 * sequence is cut into n_ranks runs of near-equal body count: every rank owns a slab of space across the world's longest
 * axis.  A re-plan moves the bodies that crossed a cut to their new owner and cuts the slabs anew once a shard is a tenth of
 * a share out of balance; the bodies stay on the devices through every plan.  Bodies keep the caller's numbering everywhere
 * in this interface.
 * One xpbd_multi_world drives this process's LOCAL shards of the n_ranks shards of the world: all of them (one process owns
 * every GPU) or one each (one process per GPU).  Every shard steps its owned bodies plus ghost copies of the remote
 * bodies within reach, and after EVERY substep the boundary bodies' 13 dynamic doubles travel in ONE all-gather (RCCL over
 * xGMI; XPBD_TRANSPORT_LOCAL = peer copies inside one process, also the single-GPU rehearsal with several shards on one
 * device).  The library builds the halo plan itself (no rank holds the global scene) and checks at the END of every frame,
 * over all ranks, how much of its travel allowance any body has used since the plan (halo_margin next to a shard boundary,
 * halo_margin + half a cell edge for a body more than two cells away from every foreign one).  Beyond it a remote contact
 * may have been missed in that frame, so the frame is UNDONE (the state it started from is kept aside on the device) and
 * either run again after a re-plan (XPBD_MULTI_AUTO_REPLAN, which also re-plans pre-emptively when another frame like the
 * last one would outrun the allowance) or reported as XPBD_E_HALO with the frame's start state in place.  Result: bit-identical to one xpbd_world over the same bodies
 * in the same order -- a state with possibly missed contacts never reaches the caller.
 *
 * Collective calls (create, upload, step, replan, download) must be made by every rank in the same order.  Failure model: a
 * rank that fails LOCALLY inside upload / replan / step / download (out of memory, a launch error) still takes part in the
 * call's collectives, each of which carries every rank's status, so EVERY rank returns an error from that call (the failing
 * rank its own, the others the same code with a message naming the rank) and a failed frame is undone everywhere.  A plan
 * (upload, replan, the re-plans of a step) that fails before any shard has been re-packed leaves the previous plan and the
 * state in place; one that fails while the shards are being re-packed leaves nothing to go back to: every later call on that
 * world fails then, destroy it.  If a collective itself cannot be enqueued the communicator is aborted (ncclCommAbort: blocked peers return with an error) and
 * every later call on that world fails: destroy it.  Argument errors are returned before any collective: the ranks' hosts
 * pass consistent arguments.  A rank that never reaches xpbd_multi_world_create leaves its peers waiting inside RCCL's
 * bootstrap; only the host's launcher can detect that.
 * Threading: xpbd_multi_world_step waits once for the broadphase's pair counts (all shards' broadphases are enqueued before
 * the first wait) and, with n_ranks > 1, for the end of the frame (the validity check).  A world with several LOCAL shards
 * enqueues their frames from one host thread per shard (started by the first step, joined by destroy): a frame is ~20 runtime
 * calls per shard and substep, more than one thread can issue for 8 GPUs in the time the GPUs need to run them.  The caller
 * still uses the world from one thread at a time.
 * The RCCL transport has run on hardware with a ONE-rank communicator only (the build box has one GPU and RCCL refuses two
 * ranks on one device); everything else is verified with XPBD_TRANSPORT_LOCAL.
 * ------------------------------------------------------------------------- */
#define XPBD_COMM_ID_BYTES 128u         /* sizeof(ncclUniqueId) */
#define XPBD_TRANSPORT_RCCL  0u
#define XPBD_TRANSPORT_LOCAL 1u         /* needs n_local == n_ranks */
#define XPBD_MULTI_AUTO_REPLAN 1u
#define XPBD_MULTI_SERIAL_ENQUEUE 4u      /* diagnostics: one host thread enqueues all local shards' frames (default with several
                                           * local shards: one enqueueing thread per shard, see Threading above) */
#define XPBD_MULTI_PLAN_THROUGH_DEVICE 2u /* diagnostics: the plan-time all-gathers go through the device transport even when
                                           * every rank lives in this process (the path a one-process-per-GPU run takes) */
#define XPBD_MULTI_FULL_PLANS 8u          /* diagnostics: every re-plan re-cuts the shards from the cell keys of the WHOLE world
                                           * (default: only the first plan and a re-plan that finds the shards a tenth of a share
                                           * out of balance do; the others keep the cuts and exchange the rims of the shards only) */

typedef struct xpbd_multi_config {
    uint32_t struct_size;   /* = sizeof(xpbd_multi_config) */
    uint32_t n_ranks;       /* shards of the whole world (<= 64) */
    uint32_t first_rank;    /* global rank of this process's first shard */
    uint32_t n_local;       /* shards driven by this process: ranks [first_rank, first_rank + n_local) */
    const int32_t *devices; /* n_local HIP device ordinals */
    uint32_t transport;     /* XPBD_TRANSPORT_* */
    uint32_t flags;         /* XPBD_MULTI_* */
    const uint8_t *comm_id; /* RCCL: XPBD_COMM_ID_BYTES from xpbd_comm_unique_id, the same on every rank */
    double   contact_pad;   /* as xpbd_world_set_contact_pad (default 0.02) */
    double   halo_margin;   /* how far a body may travel between plans (default 0.5 m) */
    uint32_t narrowphase;   /* XPBD_NARROWPHASE_* */
    uint32_t reserved;      /* must be 0 */
} xpbd_multi_config;

typedef struct xpbd_multi_world xpbd_multi_world;

/* One rank creates the communicator id; the host hands it to every rank (any channel) before xpbd_multi_world_create. */
int  xpbd_comm_unique_id(uint8_t id[XPBD_COMM_ID_BYTES]);
/* The RCCL library the collectives are bound to at run time (a copy already loaded into the process is preferred over
 * loading a second one), or NULL if none could be loaded. */
const char *xpbd_comm_library(void);
void xpbd_multi_config_default(xpbd_multi_config *cfg);
int  xpbd_multi_world_create(xpbd_multi_world **out, const xpbd_multi_config *cfg);   /* collective over all ranks (RCCL) */
void xpbd_multi_world_destroy(xpbd_multi_world *mw);
int  xpbd_multi_world_set_polytopes(xpbd_multi_world *mw, const xpbd_polytope *shapes, uint32_t n_shapes);
int  xpbd_multi_world_set_max_depenetration_speed(xpbd_multi_world *mw, double speed);   /* as xpbd_world_set_max_depenetration_speed */
/* bodies: the slice of the caller's bodies this process HANDS OVER = global indices [first_global, first_global + n_bodies) of
 * n_global (rank r hands over the r-th of n_ranks near-equal contiguous ranges, the first n_global % n_ranks one body longer;
 * any order -- which rank ends up owning a body is decided by where the body is); joints: ALL joints of the world with GLOBAL
 * body indices, the same list on every rank.  A non-finite position is XPBD_E_INVALID (the body cannot be placed in the
 * grid).  Collective: cuts the shards, moves every body to its owner and builds the halo plan. */
int  xpbd_multi_world_upload(xpbd_multi_world *mw, const xpbd_rigid *bodies, const uint32_t *shape_id, uint32_t first_global,
                             uint32_t n_bodies, uint32_t n_global, const xpbd_joint *joints, uint32_t n_joints);
/* xpbd_world_step(dt, substeps) of the whole sharded world; collective.  XPBD_E_HALO: see above (the frame was undone). */
int  xpbd_multi_world_step(xpbd_multi_world *mw, double dt, uint32_t substeps);
/* Re-cuts the shards from the bodies' current positions (re-balancing them), migrates bodies whose owner changed and
 * re-plans the halos.  Collective; clears XPBD_E_HALO. */
int  xpbd_multi_world_replan(xpbd_multi_world *mw);
int  xpbd_multi_world_synchronize(xpbd_multi_world *mw);
/* The slice this process handed over, [first_global, first_global + n), in the caller's order.  Collective (with more than
 * one process every owner's bodies travel in one all-gather: use download_owned on large worlds). */
int  xpbd_multi_world_download(xpbd_multi_world *mw, xpbd_rigid *out, uint32_t n);
/* The bodies this process's shards OWN at the moment, with their global indices (ids[k] belongs to out[k]); *n_out receives
 * their number even when it exceeds cap (then XPBD_E_CAPACITY).  Not collective. */
int  xpbd_multi_world_download_owned(xpbd_multi_world *mw, uint32_t *ids, xpbd_rigid *out, uint32_t cap, uint32_t *n_out);
/* out = {bodies of the world, owned here, ghosts here, boundary bodies here, rows per rank of the all-gather, plans made};
 * *max_displacement (optional) = the largest fraction of its travel allowance any body had used at the last check, in
 * margin-equivalent metres (x halo_margin). */
int  xpbd_multi_world_halo_stats(xpbd_multi_world *mw, uint64_t out[6], double *max_displacement);
/* out = {plans made, frames undone (halo violations), bodies that changed owner at the last plan, fewest / most bodies owned
 * by a rank, step calls, and the host time inside them in ns: enqueueing, waiting for the broadphases' pair counts, waiting
 * for the end of the frame; the host time of all plans (creation and re-plans) in ns; how many of the plans were full ones
 * (shards re-cut from the cell keys of the whole world) and how many light ones (cuts kept, only the rims exchanged)}. */
int  xpbd_multi_world_plan_stats(xpbd_multi_world *mw, uint64_t out[12]);
/* owner[g] = rank that owns body g as of the last plan (n_global entries).  COLLECTIVE when ranks live in other processes. */
int  xpbd_multi_world_owners(xpbd_multi_world *mw, uint8_t *owner, uint32_t n_global);
int  xpbd_multi_world_contact_stats(xpbd_multi_world *mw, uint64_t out[3]);             /* sums of xpbd_world_contact_stats */
/* Diagnostics, host only (no device needed), exactly as the plans compute them.  The grid cell key of a bounding-sphere
 * centre; the owner of every body from the cell keys of all bodies (the keys
 * re-packed with the longest axis of the cells' bounding box first, that sequence cut into n_ranks runs of near-equal body count, on cell boundaries unless a rank would end up more than a quarter
 * of its share off balance -- then the cell is split by body index); and one rank's halo plan from keys and owners (ascending
 * ghost ids: the remote bodies it mirrors; ascending boundary ids: its own bodies that others mirror; far (optional): per
 * owned body in ascending index, 1 if no foreign body lies within two cells, so that it may travel halo_margin + half a cell
 * edge before the halos must be re-planned, the others halo_margin). */
int64_t xpbd_halo_cell_key(const double centre[3], double cell_edge);
int  xpbd_halo_partition(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint8_t *owner);
int  xpbd_halo_plan_owned(const int64_t *cell_keys, const uint8_t *owner, uint32_t n_global, uint32_t n_ranks, uint32_t rank,
                          const xpbd_joint *joints, uint32_t n_joints, uint32_t *ghosts, uint32_t *n_ghosts, uint32_t *boundary,
                          uint32_t *n_boundary, uint8_t *far, uint32_t cap);
/* One rank's LIGHT plan, as the re-plans of xpbd_multi_world compute it: the cuts are those of xpbd_halo_partition for
 * `keys_at_cut` (where the bodies were when the slabs were cut last; every body is still held by the rank that owned it then),
 * a body's owner now (owner_now[g], all n_global of them) follows from its present cell key and those cuts, and the rank plans
 * from the bodies it holds plus the RIMS every holder would publish (bodies within two layers of a cut, bodies that change
 * owner, ends of joints that leave their holder).  own: what the rank will own (ascending); ghosts / boundary / far as
 * xpbd_halo_plan_owned, which must give the same lists for (cell_keys, owner_now). */
int  xpbd_halo_plan_light(const int64_t *keys_at_cut, const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank,
                          const xpbd_joint *joints, uint32_t n_joints, uint8_t *owner_now, uint32_t *own, uint32_t *n_own,
                          uint32_t *ghosts, uint32_t *n_ghosts, uint32_t *boundary, uint32_t *n_boundary, uint8_t *far, uint32_t cap);
/* ... the same with ownership by contiguous index ranges (rank r owns the r-th of n_ranks near-equal ranges). */
int  xpbd_halo_plan_far(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank, uint8_t *far, uint32_t cap,
                        uint32_t *n_owned);
int  xpbd_halo_plan(const int64_t *cell_keys, uint32_t n_global, uint32_t n_ranks, uint32_t rank, const xpbd_joint *joints,
                    uint32_t n_joints, uint32_t *ghosts, uint32_t *n_ghosts, uint32_t *boundary, uint32_t *n_boundary, uint32_t cap);

/* State history: the device-side counterpart of the reference app's `states: Vec<(World,
 * DebugLines)>` with its `current_state` cursor (src/app.rs:48, 206-212), which lets the user
 * scrub back through every simulated frame.  `World` is `Copy` there; here a state is the 13
 * dynamic doubles per body plus the contact masks of the last substep, copied device-to-device
 * (27 MB per state at 262 144 bodies -- thousands of frames fit in HBM).
 *   push     appends the current state, *index_out (optional) = its index (= length before the call)
 *   restore  makes state `index` the current one (neighbour lists of XPBD_MODE_CONTACTS are rebuilt
 *            by the next step / contacts_begin); stepping on from it reproduces the original run
 *            bit for bit
 *   truncate drops the states with index >= length (branching off a past state)
 * xpbd_world_upload_bodies clears the history. */
int  xpbd_world_history_push(xpbd_world *w, uint32_t *index_out);
int  xpbd_world_history_restore(xpbd_world *w, uint32_t index);
int  xpbd_world_history_truncate(xpbd_world *w, uint32_t length);
uint32_t xpbd_world_history_length(const xpbd_world *w);

/* Diagnostics: quotient[i] = a[i] / b[i], root[i] = sqrt(a[i]) computed on the
 * device with the stepper's own code generation.  Bit-exact contact lists need
 * both to be correctly rounded; the parity tests check this against the host. */
int  xpbd_selftest_div_sqrt(int32_t device, const double *a, const double *b,
                            double *quotient, double *root, uint32_t n);

/* Diagnostics: the HBM roof as this library can reach it -- a device-to-device copy of `bytes` (use far more than the
 * 256 MiB Infinity Cache) by the library's own streaming kernel, `repeats` launches timed with HIP events.
 * *gbytes_per_s = (bytes read + bytes written) / time, in 1e9 bytes per second. */
int  xpbd_selftest_hbm_copy(int32_t device, uint64_t bytes, uint32_t repeats, double *gbytes_per_s);
/* Diagnostics: what the memory system gives the ACCESS PATTERN of one substep of the pinned path -- 38 doubles read and 13
 * written per body and nothing else -- in the world's field-major layout (tile_major = 0: 51 concurrent streams of 512
 * bytes per wave) or tile-major (1: 64 bodies x all fields contiguous); best of `repeats` launches.  This, not the
 * two-stream copy above, is the roof of XPBD_MODE_PER_SUBSTEP on worlds beyond the Infinity Cache. */
int  xpbd_selftest_field_streams(int32_t device, uint64_t bodies, uint32_t tile_major, uint32_t repeats, double *gbytes_per_s);

/* Diagnostics: a GATHER of known size -- lane i reads the first read_bytes of record perm(i) of `records` records (a power of
 * two) of record_bytes each with 16-byte loads, every record exactly once, and writes one double; (read_bytes, record_bytes)
 * one of (128, 128), (64, 128), (16, 128), (192, 192), (64, 192), (128, 256): how the contact kernels read body records, mass
 * properties and manifolds.  *gbytes_per_s = (records * read_bytes + records * 8) / best time.  Under `rocprofv3 --pmc
 * FETCH_SIZE` this calibrates the counter for gathered loads (profiles/, scripts/fetch_calibration.py). */
int  xpbd_selftest_gather(int32_t device, uint32_t records, uint32_t record_bytes, uint32_t read_bytes, uint32_t repeats,
                          double *gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* XPBD_H */
