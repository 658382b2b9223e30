// host_capi.cpp -- plain-C window onto the host mirror (constraint_solver.hpp) so that
// the Python test glue and bench.py can build scenes and check the set-up math
// without a GPU.  Nothing here is on the per-substep hot path.
#include <cstring>

#include "constraint_solver.hpp"

using namespace constraint_solver;

namespace {
geometry::Polytope shape_by_code(uint32_t code, double scale)
{
    geometry::Polytope p = code == 0   ? geometry::Polytope::new_cube()
                           : code == 1 ? geometry::Polytope::new_tetrahedron()
                                       : geometry::Polytope::new_icosahedron();
    return scale == 1.0 ? p : scale * p;
}
} // namespace

extern "C" {

// Shape tables of a scene kind: verts (xyz triples) + CSR offsets.  Returns the
// number of shapes, or -1 if a capacity is too small.
int xpbdh_scene_shapes(uint32_t kind, double *verts_xyz, uint32_t cap_verts, uint32_t *offsets, uint32_t cap_shapes)
{
    const auto shapes = scene::shapes_of((scene::Kind)kind);
    uint32_t total = 0;
    for (const auto &p : shapes)
        total += (uint32_t)p.vertices.size();
    if (total > cap_verts || shapes.size() > cap_shapes)
        return -1;
    uint32_t at = 0;
    offsets[0] = 0;
    for (size_t s = 0; s < shapes.size(); ++s) {
        for (const auto &v : shapes[s].vertices) {
            verts_xyz[3 * at + 0] = v.x;
            verts_xyz[3 * at + 1] = v.y;
            verts_xyz[3 * at + 2] = v.z;
            ++at;
        }
        offsets[s + 1] = at;
    }
    return (int)shapes.size();
}

// Bodies [first, first+count) of scene (kind, seed, grid_w).
int xpbdh_scene_generate(uint32_t kind, uint64_t seed, uint32_t grid_w, uint32_t first, uint32_t count,
                         xpbd_rigid *out, uint32_t *shape_id)
{
    try {
        std::vector<rigid::Rigid> bodies;
        std::vector<uint32_t> sid;
        scene::generate((scene::Kind)kind, seed, grid_w, first, count, bodies, sid);
        if (count) {
            std::memcpy(out, bodies.data(), (size_t)count * sizeof(xpbd_rigid));
            std::memcpy(shape_id, sid.data(), (size_t)count * sizeof(uint32_t));
        }
        return XPBD_OK;
    } catch (const Error &e) {
        return e.code;
    }
}

uint32_t xpbdh_default_grid_width(uint32_t n) { return scene::default_grid_width(n); }

// Mass properties of shape `code` (0 cube, 1 tetrahedron, 2 icosahedron) scaled by
// `scale`: out = {mass, volume, com[3], inertia[9] column-major}.
void xpbdh_rigid_metrics(uint32_t code, double scale, double density, double out[14])
{
    const geometry::RigidMetrics m = shape_by_code(code, scale).rigid_metrics(density);
    out[0] = m.mass;
    out[1] = m.volume;
    std::memcpy(out + 2, &m.center_of_mass, 24);
    std::memcpy(out + 5, &m.inertia_tensor, 72);
}

// Rigid::new from those metrics; XPBD_E_SINGULAR_INERTIA mirrors the reference's panic.
int xpbdh_rigid_new(const double metrics[14], xpbd_rigid *out)
{
    geometry::RigidMetrics m;
    m.mass = metrics[0];
    m.volume = metrics[1];
    std::memcpy(&m.center_of_mass, metrics + 2, 24);
    std::memcpy(&m.inertia_tensor, metrics + 5, 72);
    try {
        const rigid::Rigid r = rigid::Rigid::make(m);
        std::memcpy(out, &r, sizeof r);
        return XPBD_OK;
    } catch (const Error &e) {
        return e.code;
    }
}

// Rigid::frame(): out = {position[3], rotation{s,x,y,z}}
void xpbdh_rigid_frame(const xpbd_rigid *r, double out[7])
{
    rigid::Rigid h;
    std::memcpy(&h, r, sizeof h);
    const Frame f = h.frame();
    std::memcpy(out, &f, 56);
}

// World::new(new_cube(), 0.5 * new_tetrahedron())  (src/world.rs:12-31, src/app.rs:29-30)
int xpbdh_world_new(xpbd_rigid *a, xpbd_rigid *b)
{
    try {
        const world::World w(geometry::Polytope::new_cube(), 0.5 * geometry::Polytope::new_tetrahedron());
        std::memcpy(a, &w.a, sizeof w.a);
        std::memcpy(b, &w.b, sizeof w.b);
        return XPBD_OK;
    } catch (const Error &e) {
        return e.code;
    }
}

// Outward plane i of a shape: out = {normal[3], displacement}; returns number of faces.
int xpbdh_shape_plane(uint32_t code, double scale, uint32_t face, double out[4])
{
    const geometry::Polytope p = shape_by_code(code, scale);
    if (face < p.faces.size()) {
        const geometry::Plane pl = p.plane(face);
        std::memcpy(out, &pl, 32);
    }
    return (int)p.faces.size();
}

// Full topology of a standard shape for xpbd_world_set_polytopes.  counts = {n_vertices, n_edges,
// n_faces, n_face_indices}; pass NULL arrays to query the counts only.
void xpbdh_polytope_arrays(uint32_t code, double scale, uint32_t counts[4], double *verts_xyz, uint32_t *edges,
                           uint32_t *face_offsets, uint32_t *face_indices, double centroid[3])
{
    const geometry::Polytope p = shape_by_code(code, scale);
    uint32_t nfi = 0;
    for (const auto &f : p.faces)
        nfi += (uint32_t)f.size();
    counts[0] = (uint32_t)p.vertices.size();
    counts[1] = (uint32_t)p.edges.size();
    counts[2] = (uint32_t)p.faces.size();
    counts[3] = nfi;
    if (!verts_xyz)
        return;
    for (size_t i = 0; i < p.vertices.size(); ++i) {
        verts_xyz[3 * i] = p.vertices[i].x;
        verts_xyz[3 * i + 1] = p.vertices[i].y;
        verts_xyz[3 * i + 2] = p.vertices[i].z;
    }
    for (size_t i = 0; i < p.edges.size(); ++i) {
        edges[2 * i] = p.edges[i][0];
        edges[2 * i + 1] = p.edges[i][1];
    }
    uint32_t at = 0;
    face_offsets[0] = 0;
    for (size_t f = 0; f < p.faces.size(); ++f) {
        for (uint32_t v : p.faces[f])
            face_indices[at++] = v;
        face_offsets[f + 1] = at;
    }
    centroid[0] = p.centroid.x;
    centroid[1] = p.centroid.y;
    centroid[2] = p.centroid.z;
}

} // extern "C"
