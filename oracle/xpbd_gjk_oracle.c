/*
 * xpbd_gjk_oracle.c -- CPU ORACLE for the GJK + EPA narrowphase of the body-body contact
 * extension.  Test infrastructure only; PARITY UNPINNED: the
 * reference has neither `gjk` nor `epa`; its only GJK-adjacent code is
 * Polytope::support / minkowski_support (src/geometry.rs:274-289), which has no caller and
 * evaluates both supports on the SAME polytope.  Kept from it: the support convention (world-space
 * vertex with the LAST maximal dot under f64::total_cmp) and the form
 * support(frames.0, d) - support(frames.1, -d), here with one polytope per frame.
 *
 * gjk:  boolean GJK (simplex of <= 4 points of A (-) B marching towards the origin).
 * epa:  expanding polytope from GJK's final tetrahedron; closest face -> penetration depth,
 *       normal (pointing from A to B) and the witness points on A and B.
 * Checked against the exact SAT of xpbd_pairs_oracle.c (tests/test_gjk_oracle.py): same
 * intersect / separated verdict, same depth to 1e-7.
 */
#include "xpbd_gjk_oracle.h"

#include <float.h>
#include <math.h>
#include <string.h>

typedef struct { o_vec3 w, a, b; uint32_t ia, ib; } mvert; /* w = a - b: a point of A (-) B with its two witnesses (and their vertex indices) */

/* index of the LAST maximal dot(v, d) under the total order (Iterator::max_by + f64::total_cmp) */
static uint32_t support_index(const o_vec3 *v, uint32_t n, o_vec3 d)
{
    uint32_t best = 0;
    double best_dot = o_dot(v[0], d);
    for (uint32_t k = 1; k < n; k++) {
        double x = o_dot(v[k], d);
        int64_t l, r;
        memcpy(&l, &best_dot, 8);
        memcpy(&r, &x, 8);
        l ^= (int64_t)((uint64_t)(l >> 63) >> 1);
        r ^= (int64_t)((uint64_t)(r >> 63) >> 1);
        if (l <= r) {
            best = k;
            best_dot = x;
        }
    }
    return best;
}

typedef struct {
    o_vec3 wa[O_MAX_VERTS], wb[O_MAX_VERTS];
    uint32_t na, nb;
} shapes_t;

static mvert minkowski_support(const shapes_t *s, o_vec3 d)
{
    mvert m;
    m.ia = support_index(s->wa, s->na, d);
    m.ib = support_index(s->wb, s->nb, o_neg(d));
    m.a = s->wa[m.ia];
    m.b = s->wb[m.ib];
    m.w = o_sub(m.a, m.b);
    return m;
}

static o_vec3 triple(o_vec3 a, o_vec3 b, o_vec3 c) { return o_cross(o_cross(a, b), c); } /* (a x b) x c */
static int same_dir(o_vec3 a, o_vec3 b) { return o_dot(a, b) > 0.0; }

/* Search direction from the segment (direction ab) towards the origin (ao = origin - a).  If the origin lies on the
 * segment's line the triple product vanishes -- two boxes stacked exactly on top of each other start like this, the
 * first two support points being (0, 0, depth) and (0, 0, depth - 2) -- and any perpendicular of ab will do: ab x the
 * coordinate axis ab has the least extent along (first minimum). */
static o_vec3 edge_direction(o_vec3 ab, o_vec3 ao)
{
    o_vec3 d = triple(ab, ao, ab);
    if (o_dot(d, d) > 0.0)
        return d;
    double ax = fabs(ab.x), ay = fabs(ab.y), az = fabs(ab.z);
    o_vec3 e = (ax <= ay && ax <= az) ? (o_vec3){ 1.0, 0.0, 0.0 } : (ay <= az ? (o_vec3){ 0.0, 1.0, 0.0 } : (o_vec3){ 0.0, 0.0, 1.0 });
    return o_cross(ab, e);
}

/* Simplex update of the boolean GJK: s[n-1] is the newest point.  Returns 1 when the origin is enclosed. */
static int do_simplex(mvert *s, uint32_t *n, o_vec3 *d)
{
    if (*n == 2) {
        o_vec3 a = s[1].w, b = s[0].w, ab = o_sub(b, a), ao = o_neg(a);
        if (same_dir(ab, ao)) {
            *d = edge_direction(ab, ao);
        } else {
            s[0] = s[1];
            *n = 1;
            *d = ao;
        }
        return 0;
    }
    if (*n == 3) {
        mvert A = s[2], B = s[1], Cc = s[0];
        o_vec3 a = A.w, ab = o_sub(B.w, a), ac = o_sub(Cc.w, a), ao = o_neg(a), abc = o_cross(ab, ac);
        if (same_dir(o_cross(abc, ac), ao)) {
            if (same_dir(ac, ao)) {
                s[0] = Cc, s[1] = A, *n = 2;
                *d = edge_direction(ac, ao);
            } else if (same_dir(ab, ao)) {
                s[0] = B, s[1] = A, *n = 2;
                *d = edge_direction(ab, ao);
            } else {
                s[0] = A, *n = 1;
                *d = ao;
            }
        } else if (same_dir(o_cross(ab, abc), ao)) {
            if (same_dir(ab, ao)) {
                s[0] = B, s[1] = A, *n = 2;
                *d = edge_direction(ab, ao);
            } else {
                s[0] = A, *n = 1;
                *d = ao;
            }
        } else if (same_dir(abc, ao)) {
            *d = abc; /* above the triangle: keep C, B, A */
        } else {
            s[0] = B, s[1] = Cc, s[2] = A; /* below: flip the winding */
            *d = o_neg(abc);
        }
        return 0;
    }
    /* tetrahedron: A newest, then B, C, D = s[2], s[1], s[0] */
    {
        mvert A = s[3], B = s[2], Cc = s[1], D = s[0];
        o_vec3 a = A.w, ao = o_neg(a);
        o_vec3 ab = o_sub(B.w, a), ac = o_sub(Cc.w, a), ad = o_sub(D.w, a);
        o_vec3 abc = o_cross(ab, ac), acd = o_cross(ac, ad), adb = o_cross(ad, ab);
        if (same_dir(abc, ao)) {
            s[0] = Cc, s[1] = B, s[2] = A, *n = 3;
            return do_simplex(s, n, d);
        }
        if (same_dir(acd, ao)) {
            s[0] = D, s[1] = Cc, s[2] = A, *n = 3;
            return do_simplex(s, n, d);
        }
        if (same_dir(adb, ao)) {
            s[0] = B, s[1] = D, s[2] = A, *n = 3;
            return do_simplex(s, n, d);
        }
        return 1;
    }
}

/* ---- EPA ------------------------------------------------------------------------------------ */
typedef struct { uint32_t i[3]; o_vec3 n; double dist; } eface;

/* Face (i0, i1, i2) of the polytope.  `opposite` = a vertex known to lie behind the face (the fourth vertex of the
 * initial tetrahedron): the winding is flipped so that the normal points away from it.  OG_NO_VERTEX = trust the
 * winding: a new face (a, b, p) over a horizon edge a -> b inherits the outward winding of the visible face the
 * edge came from.  (Orienting by the sign of the origin's distance instead breaks down exactly where it matters:
 * with the origin ON a face of the first tetrahedron -- exactly aligned boxes -- that sign is rounding noise.) */
#define OG_NO_VERTEX 0xFFFFFFFFu
static int make_face(const mvert *v, uint32_t i0, uint32_t i1, uint32_t i2, uint32_t opposite, eface *f)
{
    o_vec3 n = o_cross(o_sub(v[i1].w, v[i0].w), o_sub(v[i2].w, v[i0].w));
    double len = o_magnitude(n);
    if (!(len > 0.0))
        return 0; /* degenerate (collinear) face */
    n = o_scale(n, 1.0 / len);
    if (opposite != OG_NO_VERTEX) {
        double side = o_dot(n, o_sub(v[opposite].w, v[i0].w));
        if (!(side != 0.0))
            return 0; /* flat tetrahedron */
        if (side > 0.0) {
            uint32_t t = i1;
            i1 = i2;
            i2 = t;
            n = o_neg(n);
        }
    }
    f->i[0] = i0, f->i[1] = i1, f->i[2] = i2;
    f->n = n;
    f->dist = o_dot(n, v[i0].w);
    return 1;
}

/*
 * WARM START of the expansion (og_gjk_epa_cached): `warm` is the pair's cached direction -- the penetration normal of its last
 * query, or the direction that separated it before it came into contact.  In a settled scene the next query's answer is a
 * face of the Minkowski difference with (nearly) that normal, so the polytope is seeded with that face instead of GJK's
 * tetrahedron: four support points in directions tilted away from n = warm / |warm| by OG_WARM_TILT towards the four
 * diagonals of a tangent frame (t1 = normalize(n x e), e = the coordinate axis n has the least extent along, first minimum;
 * t2 = n x t1), in the order (+t1 +t2), (-t1 +t2), (-t1 -t2), (+t1 -t2), duplicates (same vertex pair) dropped, plus the
 * support point of -n as the apex.  With m = 3 or 4 distinct top points the polytope is the pyramid
 *     top fan (0, k, k + 1), k = 1 .. m - 2 (opposite vertex: the apex m), then sides (k + 1, k, m), k = 0 .. m - 1 (indices of
 *     the top modulo m; opposite vertex: top vertex k + 2),
 * oriented by make_face.  It is used only if every face is sound, no vertex lies more than OG_WARM_CONVEX in front of any face
 * (the five points are in convex position with exactly this face structure) and the origin is inside or on it (every face
 * distance >= 0) -- which also proves the pair penetrating, so the boolean GJK is skipped.  Otherwise GJK runs as ever.  The
 * expansion itself is unchanged, so it still ends on the globally closest face; for two boxes resting on each other it
 * ends in its first iteration instead of the seventh.
 */
#define OG_WARM_TILT   1e-3
#define OG_WARM_CONVEX 1e-10
static int seed_polytope(const shapes_t *s, o_vec3 warm, mvert *v, eface *f, uint32_t *nv_out, uint32_t *nf_out)
{
    double len = o_magnitude(warm);
    if (!(len > 0.0) || !(len <= DBL_MAX))
        return 0;
    o_vec3 n = o_scale(warm, 1.0 / len);
    double ax = fabs(n.x), ay = fabs(n.y), az = fabs(n.z);
    o_vec3 e = (ax <= ay && ax <= az) ? (o_vec3){ 1.0, 0.0, 0.0 } : (ay <= az ? (o_vec3){ 0.0, 1.0, 0.0 } : (o_vec3){ 0.0, 0.0, 1.0 });
    o_vec3 t1 = o_normalize(o_cross(n, e)), t2 = o_cross(n, t1);
    static const double sx[4] = { 1.0, -1.0, -1.0, 1.0 }, sy[4] = { 1.0, 1.0, -1.0, -1.0 };
    uint32_t m = 0;
    for (int k = 0; k < 4; k++) {
        o_vec3 d = o_add(n, o_add(o_lscale(sx[k] * OG_WARM_TILT, t1), o_lscale(sy[k] * OG_WARM_TILT, t2)));
        mvert p = minkowski_support(s, d);
        int seen = 0;
        for (uint32_t q = 0; q < m; q++)
            seen |= v[q].ia == p.ia && v[q].ib == p.ib;
        if (!seen)
            v[m++] = p;
    }
    if (m < 3)
        return 0;
    mvert apex = minkowski_support(s, o_neg(n));
    for (uint32_t q = 0; q < m; q++)
        if (v[q].ia == apex.ia && v[q].ib == apex.ib)
            return 0;
    v[m] = apex;
    uint32_t nf = 0;
    for (uint32_t k = 1; k + 1 < m; k++)
        if (!make_face(v, 0, k, k + 1, m, &f[nf++]))
            return 0;
    for (uint32_t k = 0; k < m; k++)
        if (!make_face(v, (k + 1) % m, k, m, (k + 2) % m, &f[nf++]))
            return 0;
    for (uint32_t k = 0; k < nf; k++) {
        if (!(f[k].dist >= 0.0))
            return 0; /* the origin is outside (or the face is NaN) */
        for (uint32_t q = 0; q <= m; q++)
            if (!(o_dot(f[k].n, o_sub(v[q].w, v[f[k].i[0]].w)) <= OG_WARM_CONVEX))
                return 0; /* not convex with this face structure */
    }
    *nv_out = m + 1;
    *nf_out = nf;
    return 1;
}

/* sep_dir (may be NULL): receives the direction whose support plane proved the pair separated, zero on every other exit.
 * warm (zero = none): see seed_polytope. */
static void gjk_epa(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, o_vec3 warm, o_vec3 *sep_dir, og_result *out)
{
    shapes_t s;
    if (sep_dir)
        *sep_dir = (o_vec3){ 0.0, 0.0, 0.0 };
    memset(out, 0, sizeof *out);
    out->status = OG_SEPARATED;
    if (pa->n_vertices == 0 || pb->n_vertices == 0)
        return;
    s.na = pa->n_vertices;
    s.nb = pb->n_vertices;
    for (uint32_t k = 0; k < s.na; k++)
        s.wa[k] = o_frame_mulv(fa, pa->vertices[k]);
    for (uint32_t k = 0; k < s.nb; k++)
        s.wb[k] = o_frame_mulv(fb, pb->vertices[k]);

    mvert v[OG_MAX_EPA_VERTS];
    eface f[OG_MAX_EPA_FACES];
    uint32_t nv = 4, nf = 0;
    if (!seed_polytope(&s, warm, v, f, &nv, &nf)) {
        /* ---- boolean GJK ---- */
        mvert sx[4];
        uint32_t n = 1;
        o_vec3 d = o_sub(o_frame_mulv(fb, pb->centroid), o_frame_mulv(fa, pa->centroid));
        if (!(o_dot(d, d) > 0.0))
            d = (o_vec3){ 1.0, 0.0, 0.0 };
        sx[0] = minkowski_support(&s, d);
        d = o_neg(sx[0].w);
        int hit = 0;
        for (uint32_t it = 0; it < OG_MAX_GJK_ITERS; it++) {
            out->gjk_iterations = it + 1;
            if (!(o_dot(d, d) > 0.0)) { /* origin on the simplex: touching / degenerate */
                out->status = OG_DEGENERATE;
                return;
            }
            mvert p = minkowski_support(&s, d);
            if (!(o_dot(p.w, d) > 0.0)) {
                if (sep_dir)
                    *sep_dir = d;
                return; /* the support plane does not pass the origin: separated (or just touching) */
            }
            sx[n++] = p;
            if (do_simplex(sx, &n, &d)) {
                hit = 1;
                break;
            }
        }
        if (!hit) {
            out->status = OG_DEGENERATE; /* iteration cap: grazing configuration */
            return;
        }

        /* ---- EPA from GJK's tetrahedron ---- */
        nv = 4, nf = 0;
        memcpy(v, sx, sizeof(mvert) * 4);
        static const uint32_t tet[4][3] = { {0, 1, 2}, {0, 3, 1}, {0, 2, 3}, {1, 3, 2} }; /* face k lacks vertex 3 - k */
        for (int k = 0; k < 4; k++) /* the vertex opposite face k is 3 - k */
            if (!make_face(v, tet[k][0], tet[k][1], tet[k][2], 3u - (uint32_t)k, &f[nf++])) {
                out->status = OG_DEGENERATE; /* flat tetrahedron */
                return;
            }
    }
    uint32_t best = 0;
    for (uint32_t it = 0; it < OG_MAX_EPA_ITERS; it++) {
        out->epa_iterations = it + 1;
        best = 0;
        for (uint32_t k = 1; k < nf; k++) /* first minimum */
            if (f[k].dist < f[best].dist)
                best = k;
        mvert p = minkowski_support(&s, f[best].n);
        double reach = o_dot(p.w, f[best].n);
        if (reach - f[best].dist < OG_EPA_TOLERANCE || nv == OG_MAX_EPA_VERTS)
            break;
        /* Faces that see the new point are removed and their boundary (the horizon) is re-triangulated.  A face the
         * point is coplanar with (within OG_EPA_COPLANAR) counts as seeing it: were it kept, a new point on the line
         * of one of its edges -- the Minkowski difference of two aligned boxes is a 3 x 3 x 3 grid of points -- would
         * close the horizon with a face of no area.
         * Canonical order, so that a parallel implementation reproduces it exactly: surviving faces keep
         * their relative order; horizon edges are listed by (visible face, edge) ascending, an edge a->b
         * of a visible face being on the horizon iff no other VISIBLE face holds b->a. */
        uint32_t edges[OG_MAX_EPA_FACES * 3][2], ne = 0;
        uint8_t visible[OG_MAX_EPA_FACES];
        for (uint32_t k = 0; k < nf; k++)
            visible[k] = o_dot(f[k].n, o_sub(p.w, v[f[k].i[0]].w)) > -OG_EPA_COPLANAR;
        for (uint32_t k = 0; k < nf; k++) {
            if (!visible[k])
                continue;
            for (int e = 0; e < 3; e++) {
                uint32_t ea = f[k].i[e], eb = f[k].i[(e + 1) % 3];
                int interior = 0;
                for (uint32_t q = 0; q < nf && !interior; q++) {
                    if (q == k || !visible[q])
                        continue;
                    for (int t = 0; t < 3; t++)
                        if (f[q].i[t] == eb && f[q].i[(t + 1) % 3] == ea)
                            interior = 1;
                }
                if (!interior) {
                    edges[ne][0] = ea;
                    edges[ne][1] = eb;
                    ne++;
                }
            }
        }
        uint32_t keep = 0;
        for (uint32_t k = 0; k < nf; k++)
            keep += !visible[k];
        if (ne == 0 || keep + ne > OG_MAX_EPA_FACES)
            break; /* numerical dead end or out of room: report the best face found so far */
        keep = 0;
        for (uint32_t k = 0; k < nf; k++)
            if (!visible[k])
                f[keep++] = f[k];
        nf = keep;
        v[nv] = p;
        int ok = 1;
        for (uint32_t q = 0; q < ne; q++)
            if (!make_face(v, edges[q][0], edges[q][1], nv, OG_NO_VERTEX, &f[nf++])) {
                ok = 0;
                break;
            }
        nv++;
        if (!ok) {
            out->status = OG_DEGENERATE;
            return;
        }
    }
    best = 0;
    for (uint32_t k = 1; k < nf; k++)
        if (f[k].dist < f[best].dist)
            best = k;

    /* barycentric coordinates of the origin's projection onto the closest face */
    const mvert *A = &v[f[best].i[0]], *B = &v[f[best].i[1]], *Cc = &v[f[best].i[2]];
    o_vec3 proj = o_scale(f[best].n, f[best].dist);
    o_vec3 v0 = o_sub(B->w, A->w), v1 = o_sub(Cc->w, A->w), v2 = o_sub(proj, A->w);
    double d00 = o_dot(v0, v0), d01 = o_dot(v0, v1), d11 = o_dot(v1, v1), d20 = o_dot(v2, v0), d21 = o_dot(v2, v1);
    double denom = d00 * d11 - d01 * d01;
    double bv = (d11 * d20 - d01 * d21) / denom, bw = (d00 * d21 - d01 * d20) / denom, bu = 1.0 - bv - bw;
    out->status = OG_PENETRATING;
    out->depth = f[best].dist;
    out->normal = f[best].n;
    out->point_a = o_add(o_add(o_scale(A->a, bu), o_scale(B->a, bv)), o_scale(Cc->a, bw));
    out->point_b = o_add(o_add(o_scale(A->b, bu), o_scale(B->b, bv)), o_scale(Cc->b, bw));
}

void og_gjk_epa(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, og_result *out)
{
    gjk_epa(fa, fb, pa, pb, (o_vec3){ 0.0, 0.0, 0.0 }, NULL, out);
}

/*
 * Does the support plane of direction d still separate the pair?  (The contact pipeline's cache: a pair GJK found
 * separated is tried with the direction that separated it before the query is run again -- in a settled pile nine in ten
 * separated pairs stay separated by the same plane from one substep to the next.)  The support vertices are picked in
 * each body's LOCAL space (d turned back by the conjugate rotation; last maximum under the total order), only the two
 * winners go to world space: 2 rotations + n dots instead of n rotations.
 */
int og_direction_separates(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, o_vec3 d)
{
    if (pa->n_vertices == 0 || pb->n_vertices == 0)
        return 0;
    o_vec3 da = o_qrot(o_qconj(fa.rotation), d), db = o_qrot(o_qconj(fb.rotation), o_neg(d));
    o_vec3 a = o_frame_mulv(fa, pa->vertices[support_index(pa->vertices, pa->n_vertices, da)]);
    o_vec3 b = o_frame_mulv(fb, pb->vertices[support_index(pb->vertices, pb->n_vertices, db)]);
    return !(o_dot(o_sub(a, b), d) > 0.0);
}

/* og_gjk_epa behind that cache: *axis (zero = none) is consulted first and refreshed by every full query -- with the
 * direction that separates the pair, or with the penetration normal, which WARM-STARTS the pair's next expansion
 * (seed_polytope); a degenerate query clears it. */
void og_gjk_epa_cached(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, o_vec3 *axis, og_result *out)
{
    if (o_dot(*axis, *axis) > 0.0 && og_direction_separates(fa, fb, pa, pb, *axis)) {
        memset(out, 0, sizeof *out);
        out->status = OG_SEPARATED;
        return;
    }
    gjk_epa(fa, fb, pa, pb, *axis, axis, out);
    if (out->status == OG_PENETRATING)
        *axis = out->normal; /* (measured on MI355X: trying the warm start for EVERY kept normal beats trying it only after long
                              * cold expansions -- box stacks +46 %, piles of boxes +-0, mixed polyhedra -3.5 % against no warm start;
                              * the selective rule lost on every scene, profiles/r03_k_ab_epa_warm_start.json) */
}
