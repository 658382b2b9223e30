/*
 * xpbd_gjk_oracle.h -- CPU ORACLE for the GJK + EPA narrowphase (extension).
 * Test infrastructure only; PARITY UNPINNED (the reference has no gjk / epa).  See the .c file.
 */
#ifndef XPBD_GJK_ORACLE_H
#define XPBD_GJK_ORACLE_H

#include "xpbd_oracle.h"

#ifdef __cplusplus
extern "C" {
#endif

#define OG_SEPARATED   0  /* a separating support plane was found (touching counts as separated) */
#define OG_PENETRATING 1  /* depth, normal, point_a, point_b are valid */
#define OG_DEGENERATE  2  /* origin on the simplex boundary / flat simplex / iteration cap: no answer (use SAT) */

#define OG_MAX_GJK_ITERS 32
#define OG_MAX_EPA_ITERS 48
#define OG_MAX_EPA_VERTS 52   /* 4 + OG_MAX_EPA_ITERS */
#define OG_MAX_EPA_FACES 128
#define OG_EPA_TOLERANCE 1e-10
#define OG_EPA_COPLANAR  1e-12 /* a face this close to the new point's plane is re-triangulated with it */

typedef struct {
    int32_t  status;
    uint32_t gjk_iterations, epa_iterations, reserved;
    double   depth;      /* penetration depth >= 0 */
    o_vec3   normal;     /* unit, from A towards B: moving B by depth * normal separates the pair */
    o_vec3   point_a;    /* witness point on A's surface */
    o_vec3   point_b;    /* witness point on B's surface; point_a - point_b = depth * normal */
} og_result;

void og_gjk_epa(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, og_result *out);
int  og_direction_separates(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, o_vec3 d);
void og_gjk_epa_cached(o_frame fa, o_frame fb, const o_polytope *pa, const o_polytope *pb, o_vec3 *axis, og_result *out);

#ifdef __cplusplus
}
#endif
#endif
