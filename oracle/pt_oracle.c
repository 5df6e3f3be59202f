/*
 * pt_oracle.c -- CPU ORACLE (test infrastructure, never shipped in the product path).
 * Plain-C restatement of the reference's per-pixel path-tracing loop.  PARITY UNPINNED: see pt_oracle.h
 * for what does and does not anchor it.  "ref:" citations are file:line under /root/reference.
 *
 * Compile: gcc -O2 -std=c11 -ffp-contract=off -fno-fast-math -fPIC -shared (see oracle/Makefile).
 */
#include "pt_oracle.h"

#include <math.h>
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------
 * Math (ref: Include/MathLib.h:5-152).  Every helper keeps the reference's operand order.
 * ---------------------------------------------------------------------------------------------- */
static const float PI_F = 3.14159265f;                     /* ref: MathLib.h:5 */
#define INV_PI_F (1.0f / PI_F)                              /* ref: MathLib.h:7 */
static const float NUDGE = 0.001f;                         /* ref: Main.cpp:49 */

typedef struct { float x, y, z; } v3;

static inline v3 V3(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 V3s(float s) { v3 r = { s, s, s }; return r; }
static inline v3 v3neg(v3 a) { return V3(-a.x, -a.y, -a.z); }                                 /* :80 */
static inline v3 v3add(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }            /* :82 */
static inline v3 v3sub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }            /* :83 */
static inline v3 v3mul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }            /* :84 */
static inline v3 v3muls(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }              /* :85 */
static inline v3 smulv3(float s, v3 a) { return V3(s * a.x, s * a.y, s * a.z); }              /* :86 */
static inline v3 sdivv3(float s, v3 a) { return V3(s / a.x, s / a.y, s / a.z); }              /* :88 */
static inline v3 v3cross(v3 a, v3 b) {                                                         /* :90 */
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float v3dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }           /* :91 */
static inline float v3len(v3 a) { return sqrtf(v3dot(a, a)); }                                /* :92 */
static inline v3 v3norm(v3 a) { float rcp = 1.0f / v3len(a); return v3muls(a, rcp); }         /* :93 */
/* std::min(a,b) = (b<a)?b:a ; std::max(a,b) = (a<b)?b:a */
static inline float stdmin(float a, float b) { return (b < a) ? b : a; }
static inline float stdmax(float a, float b) { return (a < b) ? b : a; }
static inline v3 v3min(v3 a, v3 b) { return V3(stdmin(a.x, b.x), stdmin(a.y, b.y), stdmin(a.z, b.z)); } /* :95 */
static inline v3 v3max(v3 a, v3 b) { return V3(stdmax(a.x, b.x), stdmax(a.y, b.y), stdmax(a.z, b.z)); } /* :96 */
static inline v3 v3lerp(v3 a, v3 b, float s) {                                                 /* :99 */
    return V3(a.x + (b.x - a.x) * s, a.y + (b.y - a.y) * s, a.z + (b.z - a.z) * s);
}
static inline float v3get(v3 a, uint32_t axis) { return axis == 0 ? a.x : (axis == 1 ? a.y : a.z); }
/* std::clamp(v, lo, hi) = (v<lo) ? lo : (hi<v) ? hi : v */
static inline float stdclamp(float v, float lo, float hi) { return (v < lo) ? lo : ((hi < v) ? hi : v); }

/* ref: MathLib.h:144-152 (no gamma, truncation, min(1,.) only; negative clamped to 0: SURVEY A-15) */
static inline uint32_t vec4_to_uint(float x, float y, float z)
{
    float fr = 255.0f * stdmin(1.0f, x), fg = 255.0f * stdmin(1.0f, y), fb = 255.0f * stdmin(1.0f, z);
    uint32_t r = (uint32_t)(uint8_t)(int32_t)(fr < 0.0f ? 0.0f : fr);
    uint32_t g = (uint32_t)(uint8_t)(int32_t)(fg < 0.0f ? 0.0f : fg);
    uint32_t b = (uint32_t)(uint8_t)(int32_t)(fb < 0.0f ? 0.0f : fb);
    return (255u << 24) + (b << 16) + (g << 8) + r;
}

/* ------------------------------------------------------------------------------------------------
 * RNG (ref: Include/Random.h:4-51; SURVEY Appendix C for the stream split)
 * ---------------------------------------------------------------------------------------------- */
uint32_t orc_wang_hash(uint32_t seed)                       /* ref: Random.h:6-13 */
{
    seed = (seed ^ 61u) ^ (seed >> 16);
    seed *= 9u; seed = seed ^ (seed >> 4);
    seed *= 0x27d4eb2du;
    seed = seed ^ (seed >> 15);
    return seed;
}
uint32_t orc_xorshift32(uint32_t* s)                        /* ref: Random.h:15-21 */
{
    *s ^= *s << 13; *s ^= *s >> 17; *s ^= *s << 5;
    return *s;
}
/* PCG-RXS-M-XS-32 ("per-lane PCG" of the GPU path): 32-bit LCG state, 32-bit permuted output */
uint32_t orc_pcg_next(uint32_t* s)
{
    uint32_t old = *s;
    *s = old * 747796405u + 2891336453u;
    uint32_t w = ((old >> ((old >> 28u) + 4u)) ^ old) * 277803737u;
    return (w >> 22u) ^ w;
}
/* stream key = (global pixel index, sample index, seed); WangHash is the reference's own (unused) hash */
uint32_t orc_pcg_seed(uint32_t pixel_index, uint32_t sample_index, uint32_t seed)
{
    uint32_t h = orc_wang_hash(seed);
    h = orc_wang_hash(h ^ sample_index);
    h = orc_wang_hash(h ^ pixel_index);
    return h;
}
float orc_u32_to_float(uint32_t u) { return (float)u * 2.3283064365387e-10f; }   /* ref: Random.h:31-34 */

typedef struct { uint32_t *m, *u, *p; int pcg; } rng_t;    /* streams M (Main.cpp), U (Util.cpp), P (Primitives.cpp) */

static inline uint32_t rng_u32(const rng_t* r, uint32_t* stream)
{
    return r->pcg ? orc_pcg_next(stream) : orc_xorshift32(stream);
}
static inline float rng_float(const rng_t* r, uint32_t* stream) { return orc_u32_to_float(rng_u32(r, stream)); }
static inline uint32_t rng_range(const rng_t* r, uint32_t* stream, uint32_t mn, uint32_t mx)  /* ref: Random.h:41-46 */
{
    if (mx - mn == 0) return mn;
    return mn + (rng_u32(r, stream) % ((mx + 1) - mn));
}

/* ------------------------------------------------------------------------------------------------
 * Scene types (ref: Primitives.h:9-115, BVH.h:28-53, Main.cpp:51-69, 94-170, 245-275)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { v3 pos, normal; } vertex_t;                /* ref: Primitives.h:9-13 */
typedef struct { vertex_t v0, v1, v2; } triangle_t;         /* ref: Primitives.h:46-51 */
typedef struct { v3 aabb_min; uint32_t left_first; v3 aabb_max; uint32_t prim_count; } bvh_node_t; /* ref: BVH.h:29-34 */

typedef struct {
    int build_option;
    bvh_node_t* nodes; uint32_t n_nodes_alloc, current_node, max_depth;
    float total_area;
    triangle_t* triangles; uint32_t n_tris;
    uint32_t* tri_indices;
    v3* centroids;
} bvh_t;

typedef struct {
    v3 origin, direction, inv_direction;
    float t;
    uint32_t obj_idx, tri_idx, bvh_depth;                   /* ref: Primitives.h:77-82 */
} ray_t;

enum { PRIM_PLANE = 0, PRIM_SPHERE = 1 };                   /* subset of Primitives.h:15-22 usable as objects */
typedef struct {
    uint32_t mat_index; int has_bvh;
    bvh_t bvh;
    int prim_type;
    v3 sphere_center; float sphere_radius, sphere_radius_sq; /* ref: Primitives.h:36-44 */
    v3 plane_normal, plane_point;                             /* ref: Primitives.h:30-34 */
} object_t;

typedef struct {
    v3 albedo; float specular, refractivity; v3 absorption; float ior; v3 emissive; float intensity; int is_light;
} material_t;

typedef struct { v3 pos, view_dir; float fov, aspect; v3 center, top_left, top_right, bottom_left; } camera_t;

struct orc_scene {
    object_t* objects; uint32_t n_objects, cap_objects;
    material_t* materials; uint32_t n_materials, cap_materials;
    uint32_t* lights; uint32_t n_lights, cap_lights;
    camera_t camera;
    int max_ray_depth, nee, cosine, rr;                     /* ref: Main.cpp:228-235 */
    int render_mode, debug_mode;
    /* render data, ref: Main.cpp:203-207 */
    uint32_t W, H; float* accumulator; uint32_t* pixels; uint32_t num_accumulated;
    orc_stats stats;
    uint32_t seed_m, seed_u, seed_p;                        /* the three per-TU s_seed copies (Appendix C) */
};

typedef struct { const orc_scene* s; rng_t rng; orc_stats st; } tctx_t;

static inline ray_t make_ray(v3 o, v3 d, float t)           /* ref: Primitives.h:61-70 */
{
    ray_t r; r.origin = o; r.direction = d; r.inv_direction = sdivv3(1.0f, d); r.t = t;
    r.obj_idx = ~0u; r.tri_idx = 0; r.bvh_depth = 0;
    return r;
}

/* ------------------------------------------------------------------------------------------------
 * Intersectors (ref: Source/Primitives.cpp:6-146)
 * ---------------------------------------------------------------------------------------------- */
static inline int intersect_triangle(v3 p0, v3 p1, v3 p2, ray_t* ray)   /* ref: Primitives.cpp:6-47 */
{
    v3 edge1 = v3sub(p1, p0);
    v3 edge2 = v3sub(p2, p0);
    v3 H = v3cross(ray->direction, edge2);
    float a = v3dot(edge1, H);
    if (fabsf(a) < 0.001f) return 0;
    float f = 1.0f / a;
    v3 S = v3sub(ray->origin, p0);
    float u = f * v3dot(S, H);
    if (u < 0.0f || u > 1.0f) return 0;
    v3 Q = v3cross(S, edge1);
    float v = f * v3dot(ray->direction, Q);
    if (v < 0.0f || u + v > 1.0f) return 0;
    float t = f * v3dot(edge2, Q);
    if (t > 0.0f && t < ray->t) { ray->t = t; return 1; }
    return 0;
}

static inline int intersect_plane(v3 normal, v3 point, ray_t* ray)      /* ref: Primitives.cpp:49-69 */
{
    float denom = v3dot(ray->direction, normal);
    if ((double)fabsf(denom) > 1e-6) {                                  /* double literal in the reference */
        v3 p0 = v3sub(point, ray->origin);
        float t = v3dot(p0, normal) / denom;
        if (t > 0.0f && t < ray->t) { ray->t = t; return 1; }
    }
    return 0;
}

static inline int intersect_sphere(v3 center, float radius_sq, ray_t* ray) /* ref: Primitives.cpp:71-114 */
{
    float t0, t1;
    v3 L = v3sub(center, ray->origin);
    float tca = v3dot(L, ray->direction);
    if (tca < 0.0f) return 0;
    float d2 = v3dot(L, L) - tca * tca;
    if (d2 > radius_sq) return 0;
    float thc = sqrtf(radius_sq - d2);
    t0 = tca - thc;
    t1 = tca + thc;
    if (t0 > t1) { float tmp = t0; t0 = t1; t1 = tmp; }
    if (t0 < 0.0f) { t0 = t1; if (t0 < 0.0f) return 0; }
    if (t0 < ray->t) { ray->t = t0; return 1; }
    return 0;
}

/* ref: Primitives.cpp:116-130.  _mm_max_ps(a,b) = a>b?a:b ; _mm_min_ps(a,b) = a<b?a:b (second operand on NaN);
 * then std::min/std::max over the three lanes (SURVEY A-18). */
static inline float intersect_aabb_sse(v3 bmin, v3 bmax, const ray_t* ray)
{
    float t1x = (bmin.x - ray->origin.x) * ray->inv_direction.x, t2x = (bmax.x - ray->origin.x) * ray->inv_direction.x;
    float t1y = (bmin.y - ray->origin.y) * ray->inv_direction.y, t2y = (bmax.y - ray->origin.y) * ray->inv_direction.y;
    float t1z = (bmin.z - ray->origin.z) * ray->inv_direction.z, t2z = (bmax.z - ray->origin.z) * ray->inv_direction.z;
    float vmaxx = t1x > t2x ? t1x : t2x, vminx = t1x < t2x ? t1x : t2x;
    float vmaxy = t1y > t2y ? t1y : t2y, vminy = t1y < t2y ? t1y : t2y;
    float vmaxz = t1z > t2z ? t1z : t2z, vminz = t1z < t2z ? t1z : t2z;
    float tmax = stdmin(vmaxx, stdmin(vmaxy, vmaxz));
    float tmin = stdmax(vminx, stdmax(vminy, vminz));
    if (tmax >= tmin && tmin < ray->t && tmax > 0.0f) return tmin;
    return 1e30f;
}

/* ------------------------------------------------------------------------------------------------
 * BVH build (ref: Source/BVH.cpp:11-59, 188-366)
 * ---------------------------------------------------------------------------------------------- */
static float triangle_area(const triangle_t* t)             /* ref: Primitives.cpp:270-278 */
{
    float a = v3len(v3sub(t->v1.pos, t->v0.pos));
    float b = v3len(v3sub(t->v2.pos, t->v0.pos));
    float c = v3len(v3sub(t->v2.pos, t->v1.pos));
    float s = (a + b + c) / 2.0f;
    return sqrtf(s * (s - a) * (s - b) * (s - c));
}
static float aabb_half_area(v3 mn, v3 mx)                   /* ref: Primitives.cpp:280-284 ("GetAABBVolume") */
{
    v3 e = v3sub(mx, mn);
    return e.x * e.y + e.y * e.z + e.z * e.x;
}
static void grow_aabb(v3* mn, v3* mx, v3 p) { *mn = v3min(*mn, p); *mx = v3max(*mx, p); } /* ref: Primitives.cpp:286-290 */

static void calc_node_bounds(bvh_t* b, bvh_node_t* node)    /* ref: BVH.cpp:188-202 */
{
    node->aabb_min = V3s(1e30f);
    node->aabb_max = V3s(-1e30f);
    for (uint32_t i = node->left_first; i < node->left_first + node->prim_count; ++i) {
        const triangle_t* tri = &b->triangles[b->tri_indices[i]];
        /* TriangleBounds, ref: Primitives.cpp:232-243 */
        v3 pmin = tri->v0.pos, pmax = tri->v0.pos;
        pmin = v3min(pmin, tri->v1.pos); pmax = v3max(pmax, tri->v1.pos);
        pmin = v3min(pmin, tri->v2.pos); pmax = v3max(pmax, tri->v2.pos);
        node->aabb_min = v3min(node->aabb_min, pmin);
        node->aabb_max = v3max(node->aabb_max, pmax);
    }
}

static float evaluate_sah(const bvh_t* b, const bvh_node_t* node, uint32_t axis, float split_pos) /* ref: BVH.cpp:299-327 */
{
    v3 lmin = V3s(1e30f), lmax = V3s(-1e30f), rmin = V3s(1e30f), rmax = V3s(-1e30f);
    uint32_t left_count = 0, right_count = 0;
    for (uint32_t i = node->left_first; i < node->left_first + node->prim_count; ++i) {
        const triangle_t* tri = &b->triangles[b->tri_indices[i]];
        v3 c = b->centroids[b->tri_indices[i]];
        if (v3get(c, axis) < split_pos) {
            left_count++;
            grow_aabb(&lmin, &lmax, tri->v0.pos); grow_aabb(&lmin, &lmax, tri->v1.pos); grow_aabb(&lmin, &lmax, tri->v2.pos);
        } else {
            right_count++;
            grow_aabb(&rmin, &rmax, tri->v0.pos); grow_aabb(&rmin, &rmax, tri->v1.pos); grow_aabb(&rmin, &rmax, tri->v2.pos);
        }
    }
    /* empty side: extent -2e30 -> area +inf -> 0*inf = NaN -> the "<" test at the call site rejects it */
    return (float)left_count * aabb_half_area(lmin, lmax) + (float)right_count * aabb_half_area(rmin, rmax);
}

static void subdivide(bvh_t* b, uint32_t node_index, uint32_t depth);

static void split_node(bvh_t* b, bvh_node_t* node, uint32_t axis, float split_pos, uint32_t depth) /* ref: BVH.cpp:329-366 */
{
    int32_t i = (int32_t)node->left_first;
    int32_t j = i + (int32_t)node->prim_count - 1;
    while (i <= j) {
        if (v3get(b->centroids[b->tri_indices[i]], axis) < split_pos) {
            i++;
        } else {
            uint32_t tmp = b->tri_indices[i]; b->tri_indices[i] = b->tri_indices[j]; b->tri_indices[j] = tmp; j--;
        }
    }
    uint32_t left_count = (uint32_t)i - node->left_first;
    if (left_count == 0 || left_count == node->prim_count) return;

    uint32_t left_child = b->current_node++;
    uint32_t right_child = b->current_node++;
    b->nodes[left_child].left_first = node->left_first;
    b->nodes[left_child].prim_count = left_count;
    b->nodes[right_child].left_first = (uint32_t)i;
    b->nodes[right_child].prim_count = node->prim_count - left_count;
    node->left_first = left_child;
    node->prim_count = 0;
    calc_node_bounds(b, &b->nodes[left_child]);
    calc_node_bounds(b, &b->nodes[right_child]);
    subdivide(b, left_child, depth + 1);
    subdivide(b, right_child, depth + 1);
}

static void subdivide(bvh_t* b, uint32_t node_index, uint32_t depth)   /* ref: BVH.cpp:204-297 */
{
    if (depth > b->max_depth) b->max_depth = depth;
    bvh_node_t* node = &b->nodes[node_index];

    if (b->build_option == ORC_BUILD_NAIVE) {                          /* ref: BVH.cpp:208-224 */
        if (node->prim_count <= 2) return;
        v3 extent = v3sub(node->aabb_max, node->aabb_min);
        uint32_t axis = 0;
        if (extent.y > extent.x) axis = 1;
        if (extent.z > v3get(extent, axis)) axis = 2;
        float split_pos = v3get(node->aabb_min, axis) + v3get(extent, axis) * 0.5f;
        split_node(b, node, axis, split_pos, depth);
    } else if (b->build_option == ORC_BUILD_SAH_INTERVALS) {           /* ref: BVH.cpp:225-259 */
        float parent_cost = aabb_half_area(node->aabb_min, node->aabb_max) * (float)node->prim_count;
        float cheapest_cost = 1e30f; uint32_t cheapest_axis = 0; float cheapest_pos = 0.0f;
        for (uint32_t split_idx = 0; split_idx < 8; ++split_idx) {
            for (uint32_t axis = 0; axis < 3; ++axis) {
                float axis_width = v3get(node->aabb_max, axis) - v3get(node->aabb_min, axis);
                float split_pos = axis_width * ((float)split_idx / 8) + v3get(node->aabb_min, axis);
                float split_cost = evaluate_sah(b, node, axis, split_pos);
                if (split_cost < cheapest_cost) { cheapest_cost = split_cost; cheapest_axis = axis; cheapest_pos = split_pos; }
            }
        }
        if (cheapest_cost >= parent_cost) return;
        split_node(b, node, cheapest_axis, cheapest_pos, depth);
    } else if (b->build_option == ORC_BUILD_SAH_PRIMITIVES) {          /* ref: BVH.cpp:260-296 */
        float parent_cost = aabb_half_area(node->aabb_min, node->aabb_max) * (float)node->prim_count;
        float cheapest_cost = 1e30f; uint32_t cheapest_axis = 0; float cheapest_pos = 0.0f;
        for (uint32_t i = node->left_first; i < node->left_first + node->prim_count; ++i) {
            v3 c = b->centroids[b->tri_indices[i]];
            for (uint32_t axis = 0; axis < 3; ++axis) {
                float split_pos = v3get(c, axis);
                float split_cost = evaluate_sah(b, node, axis, split_pos);
                /* the reference never updates cheapest_cost here (SURVEY A-5): the node never splits */
                if (split_cost < cheapest_cost) { cheapest_axis = axis; cheapest_pos = split_pos; }
            }
        }
        if (cheapest_cost >= parent_cost) return;
        split_node(b, node, cheapest_axis, cheapest_pos, depth);
    }
}

static void bvh_root_and_subdivide(bvh_t* b)
{
    bvh_node_t* root = &b->nodes[b->current_node++];
    root->left_first = 0;
    root->prim_count = b->n_tris;
    calc_node_bounds(b, root);
    subdivide(b, 0, 0);
}

static int bvh_build(bvh_t* b, const float* verts, uint32_t nverts, const uint32_t* indices, uint32_t nidx, int opt) /* ref: BVH.cpp:11-45 */
{
    memset(b, 0, sizeof(*b));
    b->build_option = opt;
    b->n_tris = nidx / 3;
    if (b->n_tris == 0) return -1;
    b->triangles = (triangle_t*)malloc(sizeof(triangle_t) * b->n_tris);
    b->tri_indices = (uint32_t*)malloc(sizeof(uint32_t) * b->n_tris);
    b->centroids = (v3*)malloc(sizeof(v3) * b->n_tris);
    for (uint32_t i = 0, k = 0; i < b->n_tris; ++i, k += 3) {
        uint32_t ia = indices[k], ib = indices[k + 1], ic = indices[k + 2];
        if (ia >= nverts || ib >= nverts || ic >= nverts) return -2;
        const float* a = verts + 6 * (size_t)ia; const float* bb = verts + 6 * (size_t)ib; const float* c = verts + 6 * (size_t)ic;
        b->triangles[i].v0.pos = V3(a[0], a[1], a[2]);   b->triangles[i].v0.normal = V3(a[3], a[4], a[5]);
        b->triangles[i].v1.pos = V3(bb[0], bb[1], bb[2]); b->triangles[i].v1.normal = V3(bb[3], bb[4], bb[5]);
        b->triangles[i].v2.pos = V3(c[0], c[1], c[2]);   b->triangles[i].v2.normal = V3(c[3], c[4], c[5]);
        b->total_area += triangle_area(&b->triangles[i]);
    }
    for (uint32_t i = 0; i < b->n_tris; ++i) b->tri_indices[i] = i;
    for (uint32_t i = 0; i < b->n_tris; ++i) {                          /* TriangleCentroid, ref: Primitives.cpp:255-258 */
        const triangle_t* t = &b->triangles[i];
        b->centroids[i] = v3muls(v3add(v3add(t->v0.pos, t->v1.pos), t->v2.pos), 0.3333f);
    }
    b->n_nodes_alloc = b->n_tris * 2 - 1;
    b->nodes = (bvh_node_t*)calloc(b->n_nodes_alloc, sizeof(bvh_node_t));
    bvh_root_and_subdivide(b);
    return 0;
}

static void bvh_rebuild(bvh_t* b, int opt)                  /* ref: BVH.cpp:47-59 (tri_indices keep their order) */
{
    b->build_option = opt;
    b->current_node = 0;
    b->max_depth = 0;
    bvh_root_and_subdivide(b);
}

static void bvh_free(bvh_t* b) { free(b->nodes); free(b->triangles); free(b->tri_indices); free(b->centroids); }

/* ------------------------------------------------------------------------------------------------
 * Traversal (ref: Source/BVH.cpp:61-127) and scene intersection (ref: Source/Main.cpp:299-316)
 * ---------------------------------------------------------------------------------------------- */
static int bvh_traverse(const bvh_t* b, ray_t* ray, orc_stats* st)
{
    int result = 0;
    const bvh_node_t* node = &b->nodes[0];
    const bvh_node_t* stack[64];
    uint32_t stack_ptr = 0;
    for (;;) {
        if (node->prim_count > 0) {
            for (uint32_t i = node->left_first; i < node->left_first + node->prim_count; ++i) {
                const triangle_t* tri = &b->triangles[b->tri_indices[i]];
                st->tri_tests++;
                if (intersect_triangle(tri->v0.pos, tri->v1.pos, tri->v2.pos, ray)) {
                    ray->tri_idx = b->tri_indices[i];
                    result = 1;
                }
            }
            if (stack_ptr == 0) break;
            node = stack[--stack_ptr];
            continue;
        }
        const bvh_node_t* left = &b->nodes[node->left_first];
        const bvh_node_t* right = &b->nodes[node->left_first + 1];
        st->inner_steps++;
        float left_dist = intersect_aabb_sse(left->aabb_min, left->aabb_max, ray);
        float right_dist = intersect_aabb_sse(right->aabb_min, right->aabb_max, ray);
        if (left_dist > right_dist) {
            float td = left_dist; left_dist = right_dist; right_dist = td;
            const bvh_node_t* tn = left; left = right; right = tn;
        }
        if (left_dist == 1e30f) {
            if (stack_ptr == 0) break;
            node = stack[--stack_ptr];
        } else {
            ray->bvh_depth++;
            st->bvh_depth_sum++;
            node = left;
            if (right_dist != 1e30f) stack[stack_ptr++] = right;
        }
    }
    return result;
}

static void intersect_scene(tctx_t* c, ray_t* ray)          /* ref: Main.cpp:299-316 */
{
    const orc_scene* s = c->s;
    c->st.traced_rays++;
    for (uint32_t obj_idx = 0; obj_idx < s->n_objects; ++obj_idx) {
        const object_t* obj = &s->objects[obj_idx];
        int hit;
        if (obj->has_bvh) hit = bvh_traverse(&obj->bvh, ray, &c->st);
        else if (obj->prim_type == PRIM_SPHERE) hit = intersect_sphere(obj->sphere_center, obj->sphere_radius_sq, ray);
        else hit = intersect_plane(obj->plane_normal, obj->plane_point, ray);
        if (hit) ray->obj_idx = obj_idx;
    }
}

typedef struct { v3 pos, normal; const material_t* mat; } hit_t;

static hit_t get_hit_result(tctx_t* c, const ray_t* ray)    /* ref: Main.cpp:325-338 */
{
    const orc_scene* s = c->s;
    hit_t h;
    h.pos = v3add(ray->origin, v3muls(ray->direction, ray->t));
    const object_t* obj = &s->objects[ray->obj_idx];
    if (obj->has_bvh) {
        h.normal = obj->bvh.triangles[ray->tri_idx].v0.normal;          /* TriangleNormal, ref: Primitives.cpp:148-151 */
        c->st.closest_hits++;
    } else if (obj->prim_type == PRIM_SPHERE) {
        h.normal = v3norm(v3sub(h.pos, obj->sphere_center));            /* SphereNormal, ref: Primitives.cpp:153-156 */
    } else {
        h.normal = obj->plane_normal;                                    /* PlaneNormal, ref: Primitives.cpp:158-161 */
    }
    h.mat = &s->materials[obj->mat_index];
    return h;
}

/* ------------------------------------------------------------------------------------------------
 * Sampling / optics (ref: Source/Util.cpp:7-54)
 * ---------------------------------------------------------------------------------------------- */
static v3 ball_sample(tctx_t* c, uint32_t* stream)          /* rejection loop of Util.cpp:10-13 / :24-27; draws x,y,z in that order */
{
    v3 dir;
    do {
        float x = rng_float(&c->rng, stream) * 2.0f - 1.0f;
        float y = rng_float(&c->rng, stream) * 2.0f - 1.0f;
        float z = rng_float(&c->rng, stream) * 2.0f - 1.0f;
        dir = V3(x, y, z);
    } while (v3dot(dir, dir) > 1.0f);
    return dir;
}
static v3 uniform_hemisphere_sample(tctx_t* c, v3 normal)   /* ref: Util.cpp:7-19 */
{
    v3 dir = ball_sample(c, c->rng.u);
    if (v3dot(dir, normal) < 0.0f) dir = v3mul(dir, V3s(-1.0f));
    return v3norm(dir);
}
static v3 cosine_weighted_diffuse_reflection(tctx_t* c, v3 normal) /* ref: Util.cpp:21-30 */
{
    v3 dir = ball_sample(c, c->rng.u);
    return v3norm(v3add(normal, v3norm(dir)));
}
static float survival_probability_rr(v3 albedo)             /* ref: Util.cpp:32-35 */
{
    return stdclamp(stdmax(stdmax(albedo.x, albedo.y), albedo.z), 0.1f, 1.0f);
}
static v3 reflect_dir(v3 dir, v3 normal)                    /* ref: Util.cpp:37-40 */
{
    return v3sub(dir, v3muls(smulv3(2.0f, normal), v3dot(dir, normal)));
}
static float fresnel(float in, float out, float ior_outside, float ior_inside) /* ref: Util.cpp:42-49 */
{
    float sPolarized = (ior_outside * in - ior_inside * out) / (ior_outside * in + ior_inside * out);
    float pPolarized = (ior_outside * out - ior_inside * in) / (ior_outside * out + ior_inside * in);
    return 0.5f * ((sPolarized * sPolarized) + (pPolarized * pPolarized));
}
static v3 refract_dir(v3 dir, v3 normal, float eta, float cosi, float k) /* ref: Util.cpp:51-54 */
{
    return v3norm(v3add(v3muls(dir, eta), smulv3(eta * cosi - sqrtf(k), normal)));
}

/* ------------------------------------------------------------------------------------------------
 * Light sampling (ref: Source/Main.cpp:340-394)
 * ---------------------------------------------------------------------------------------------- */
typedef struct { v3 pos, to_light; float distance; v3 normal, emission; float area; uint32_t obj_idx; } light_sample_t;

static light_sample_t get_random_light_sample(tctx_t* c, v3 hit_pos)
{
    const orc_scene* s = c->s;
    light_sample_t ls; memset(&ls, 0, sizeof(ls)); ls.obj_idx = ~0u;
    if (s->n_lights > 0) {
        ls.obj_idx = s->lights[rng_range(&c->rng, c->rng.m, 0u, s->n_lights - 1)];
        const object_t* light = &s->objects[ls.obj_idx];
        if (light->has_bvh) {                                            /* ref: Main.cpp:360-368 */
            const triangle_t* tri = &light->bvh.triangles[rng_range(&c->rng, c->rng.m, 0u, light->bvh.n_tris - 1)];
            /* RandomPointTriangle, ref: Primitives.cpp:170-186 (stream P) */
            float u0 = rng_float(&c->rng, c->rng.p);
            float u1 = rng_float(&c->rng, c->rng.p);
            float alpha = u0, beta = u1;
            if (alpha + beta > 1.0f) { alpha = 1.0f - alpha; beta = 1.0f - beta; }
            float gamma = 1.0f - beta - alpha;
            ls.pos = v3add(v3add(smulv3(alpha, tri->v0.pos), smulv3(beta, tri->v1.pos)), smulv3(gamma, tri->v2.pos));
            ls.normal = tri->v0.normal;
            ls.area = light->bvh.total_area / 2.0f;
        } else {                                                         /* sphere only, ref: Main.cpp:371-384 */
            /* RandomPointSphereFacing, ref: Primitives.cpp:214-220 */
            v3 to_pos = v3norm(v3sub(hit_pos, light->sphere_center));
            v3 dir = uniform_hemisphere_sample(c, to_pos);
            ls.pos = v3add(light->sphere_center, smulv3(light->sphere_radius, dir));
            ls.normal = v3norm(v3sub(ls.pos, light->sphere_center));
            ls.area = 2.0f * PI_F * light->sphere_radius_sq;
        }
        ls.to_light = v3sub(ls.pos, hit_pos);
        ls.distance = v3len(ls.to_light);
        ls.to_light = v3norm(ls.to_light);
        const material_t* lm = &s->materials[light->mat_index];
        ls.emission = v3muls(lm->emissive, lm->intensity);
    }
    return ls;
}

/* ------------------------------------------------------------------------------------------------
 * Integrators
 * ---------------------------------------------------------------------------------------------- */
static v3 trace_path_advanced(tctx_t* c, ray_t* ray)        /* ref: Main.cpp:396-579 */
{
    const orc_scene* s = c->s;
    v3 throughput = V3s(1.0f), energy = V3s(0.0f);
    uint8_t ray_depth = 0;
    int is_specular_ray = 0;

    while ((int)ray_depth <= s->max_ray_depth) {
        intersect_scene(c, ray);

        if (ray_depth == 0 && s->debug_mode == ORC_DEBUG_BVH_DEPTH) {
            energy = v3add(energy, v3lerp(V3(0.0f, 1.0f, 0.0f), V3(1.0f, 0.0f, 0.0f), (float)ray->bvh_depth / 30.0f));
            break;
        }
        if (ray->obj_idx == ~0u) break;

        hit_t hit = get_hit_result(c, ray);

        if (hit.mat->is_light) {
            if (!s->nee || ray_depth == 0 || is_specular_ray)
                energy = v3add(energy, v3muls(v3mul(throughput, hit.mat->emissive), hit.mat->intensity));
            break;
        }

        v3 brdf_diffuse = v3muls(hit.mat->albedo, INV_PI_F);
        float diffuse_weight = stdmax(0.0f, 1.0f - hit.mat->specular - hit.mat->refractivity);

        if (s->n_lights > 0 && s->nee && diffuse_weight > 0.001f) {
            light_sample_t ls = get_random_light_sample(c, hit.pos);
            float NdotL = v3dot(hit.normal, ls.to_light);
            float NLdotL = v3dot(ls.normal, v3neg(ls.to_light));
            if (NdotL > 0.0f && NLdotL > 0.0f) {
                ray_t shadow = make_ray(v3add(hit.pos, v3muls(ls.to_light, NUDGE)), ls.to_light, ls.distance - 2.0f * NUDGE);
                intersect_scene(c, &shadow);
                int occluded = shadow.obj_idx != ~0u;
                if (!occluded) {
                    float solid_angle = (NLdotL * ls.area) / (ls.distance * ls.distance);
                    float light_pdf = 1.0f / solid_angle;
                    v3 e = v3muls(throughput, NdotL / light_pdf);
                    e = v3mul(e, brdf_diffuse);
                    e = v3mul(e, ls.emission);
                    e = v3muls(e, (float)s->n_lights);
                    e = v3muls(e, diffuse_weight);
                    energy = v3add(energy, e);
                }
            }
        }

        if (s->rr) {
            float p = survival_probability_rr(hit.mat->albedo);
            if (p < rng_float(&c->rng, c->rng.m)) break;
            else throughput = v3mul(throughput, V3s(1.0f / p));
        }

        float r = rng_float(&c->rng, c->rng.m);

        if (r < hit.mat->specular) {                                     /* ref: Main.cpp:480-487 */
            v3 sd = reflect_dir(ray->direction, hit.normal);
            *ray = make_ray(v3add(hit.pos, v3muls(sd, NUDGE)), sd, 1e34f);
            throughput = v3mul(throughput, hit.mat->albedo);
            is_specular_ray = 1;
        } else if (r < hit.mat->specular + hit.mat->refractivity) {      /* ref: Main.cpp:488-546 */
            v3 N = hit.normal;
            float cosi = stdclamp(v3dot(N, ray->direction), -1.0f, 1.0f);
            float etai = 1.0f, etat = hit.mat->ior;
            float Fr = 1.0f;
            int inside = 1;
            if (cosi < 0.0f) { cosi = -cosi; inside = 0; }
            else { float tmp = etai; etai = etat; etat = tmp; N = v3neg(N); }
            float eta = etai / etat;
            float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
            if (k >= 0.0f) {
                v3 rd = refract_dir(ray->direction, N, eta, cosi, k);
                float angle_in = v3dot(ray->direction, hit.normal);
                float angle_out = v3dot(rd, hit.normal);
                Fr = fresnel(angle_in, angle_out, etai, etat);
                if (rng_float(&c->rng, c->rng.m) > Fr) {
                    throughput = v3mul(throughput, hit.mat->albedo);
                    if (inside) {
                        v3 ab;
                        ab.x = expf(-hit.mat->absorption.x * ray->t);
                        ab.y = expf(-hit.mat->absorption.y * ray->t);
                        ab.z = expf(-hit.mat->absorption.z * ray->t);
                        throughput = v3mul(throughput, ab);
                    }
                    *ray = make_ray(v3add(hit.pos, v3muls(rd, NUDGE)), rd, 1e34f);
                    is_specular_ray = 1;
                } else {
                    v3 sd = reflect_dir(ray->direction, hit.normal);
                    *ray = make_ray(v3add(hit.pos, v3muls(sd, NUDGE)), sd, 1e34f);
                    throughput = v3mul(throughput, hit.mat->albedo);
                    is_specular_ray = 1;
                }
            }
            /* k < 0 (TIR): ray left untouched, re-intersected next iteration (SURVEY A-3) */
        } else {                                                         /* ref: Main.cpp:547-570 */
            v3 dd; float NdotR, pdf;
            if (s->cosine) {
                dd = cosine_weighted_diffuse_reflection(c, hit.normal);
                NdotR = v3dot(dd, hit.normal);
                pdf = 1.0f / (2.0f * PI_F);
            } else {
                dd = uniform_hemisphere_sample(c, hit.normal);
                NdotR = v3dot(dd, hit.normal);
                pdf = NdotR / PI_F;
            }
            *ray = make_ray(v3add(hit.pos, v3muls(dd, NUDGE)), dd, 1e34f);
            throughput = v3mul(throughput, smulv3(NdotR / pdf, brdf_diffuse));
            is_specular_ray = 0;
        }
        ray_depth++;
    }

    if (s->debug_mode == ORC_DEBUG_RAY_DEPTH)
        energy = v3lerp(V3(0.0f, 1.0f, 0.0f), V3(1.0f, 0.0f, 0.0f), (float)ray_depth / (float)s->max_ray_depth);
    return energy;
}

static v3 trace_path_brute(tctx_t* c, ray_t* ray, uint8_t ray_depth) /* ref: Main.cpp:581-689 */
{
    const orc_scene* s = c->s;
    v3 final_color = V3s(0.0f);
    if ((int)ray_depth > s->max_ray_depth) return final_color;

    intersect_scene(c, ray);

    if (ray_depth == 0 && s->debug_mode == ORC_DEBUG_BVH_DEPTH)
        return v3lerp(V3(0.0f, 1.0f, 0.0f), V3(1.0f, 0.0f, 0.0f), (float)ray->bvh_depth / 30.0f);
    if (ray->obj_idx == ~0u) return final_color;

    hit_t hit = get_hit_result(c, ray);
    if (hit.mat->is_light) return v3muls(hit.mat->emissive, hit.mat->intensity);

    float r = rng_float(&c->rng, c->rng.m);

    if (r < hit.mat->specular) {
        v3 sd = reflect_dir(ray->direction, hit.normal);
        ray_t nr = make_ray(v3add(hit.pos, v3muls(sd, NUDGE)), sd, 1e34f);
        final_color = v3add(final_color, v3mul(hit.mat->albedo, trace_path_brute(c, &nr, (uint8_t)(ray_depth + 1))));
    } else if (r < hit.mat->specular + hit.mat->refractivity) {
        v3 N = hit.normal;
        float cosi = stdclamp(v3dot(N, ray->direction), -1.0f, 1.0f);
        float etai = 1.0f, etat = hit.mat->ior;
        float Fr = 1.0f;
        int inside = 1;
        if (cosi < 0.0f) { cosi = -cosi; inside = 0; }
        else { float tmp = etai; etai = etat; etat = tmp; N = v3neg(N); }
        float eta = etai / etat;
        float k = 1.0f - eta * eta * (1.0f - cosi * cosi);
        if (k >= 0.0f) {
            v3 rd = refract_dir(ray->direction, N, eta, cosi, k);
            ray_t rr = make_ray(v3add(hit.pos, v3muls(rd, NUDGE)), rd, 1e34f);
            float angle_in = v3dot(ray->direction, hit.normal);
            float angle_out = v3dot(rd, hit.normal);
            Fr = fresnel(angle_in, angle_out, etai, etat);
            if (rng_float(&c->rng, c->rng.m) > Fr) {
                final_color = v3add(final_color, v3mul(hit.mat->albedo, trace_path_brute(c, &rr, (uint8_t)(ray_depth + 1))));
                if (inside) {
                    v3 ab;
                    ab.x = expf(-hit.mat->absorption.x * ray->t);
                    ab.y = expf(-hit.mat->absorption.y * ray->t);
                    ab.z = expf(-hit.mat->absorption.z * ray->t);
                    final_color = v3mul(final_color, ab);
                }
            } else {
                v3 sd = reflect_dir(ray->direction, hit.normal);
                ray_t nr = make_ray(v3add(hit.pos, v3muls(sd, NUDGE)), sd, 1e34f);
                final_color = v3add(final_color, v3mul(hit.mat->albedo, trace_path_brute(c, &nr, (uint8_t)(ray_depth + 1))));
            }
        }
    } else {
        v3 dd = uniform_hemisphere_sample(c, hit.normal);
        ray_t nr = make_ray(v3add(hit.pos, v3muls(dd, NUDGE)), dd, 1e34f);
        float cosi = v3dot(dd, hit.normal);
        v3 irradiance = smulv3(cosi, trace_path_brute(c, &nr, (uint8_t)(ray_depth + 1)));
        v3 diffuse_brdf = v3muls(hit.mat->albedo, INV_PI_F);
        final_color = v3add(final_color, v3mul(smulv3(2.0f * PI_F, diffuse_brdf), irradiance));
    }
    return final_color;
}

/* ------------------------------------------------------------------------------------------------
 * Camera (ref: Main.cpp:98-102, 133-149)
 * ---------------------------------------------------------------------------------------------- */
static void update_screen_plane(camera_t* cam)              /* ref: Main.cpp:143-149 */
{
    cam->center = v3add(cam->pos, smulv3(cam->fov, cam->view_dir));
    cam->top_left = v3add(cam->center, V3(-cam->aspect, 1.0f, 0.0f));
    cam->top_right = v3add(cam->center, V3(cam->aspect, 1.0f, 0.0f));
    cam->bottom_left = v3add(cam->center, V3(-cam->aspect, -1.0f, 0.0f));
}
static ray_t camera_get_ray(const camera_t* cam, float u, float v) /* ref: Main.cpp:133-140 */
{
    v3 pixel_pos = v3add(v3add(cam->top_left, smulv3(u, v3sub(cam->top_right, cam->top_left))),
                         smulv3(v, v3sub(cam->bottom_left, cam->top_left)));
    return make_ray(cam->pos, v3norm(v3sub(pixel_pos, cam->pos)), 1e34f);
}

/* ------------------------------------------------------------------------------------------------
 * Frame driver (ref: Main.cpp:691-755)
 * ---------------------------------------------------------------------------------------------- */
static void shade_pixel(tctx_t* c, orc_scene* s, uint32_t px, uint32_t py, float inv_w, float inv_h,
                        int rng_mode, uint32_t seed, uint32_t* pcg_state)
{
    const float screen_u = (float)px * inv_w;                            /* ref: Main.cpp:713-714 */
    const float screen_v = (float)py * inv_h;
    uint32_t pos = py * s->W + px;
    if (rng_mode == ORC_RNG_PIXEL_PCG) *pcg_state = orc_pcg_seed(pos, s->num_accumulated - 1, seed);

    ray_t ray = camera_get_ray(&s->camera, screen_u, screen_v);
    v3 col = V3s(0.0f);
    if (s->render_mode == ORC_MODE_COMPARISON) {                         /* ref: Main.cpp:719-733 */
        if (px < (s->W / 2)) col = trace_path_brute(c, &ray, 0);
        else col = trace_path_advanced(c, &ray);
    } else if (s->render_mode == ORC_MODE_BRUTE_FORCE) {
        col = trace_path_brute(c, &ray, 0);
    } else if (s->render_mode == ORC_MODE_ADVANCED) {
        col = trace_path_advanced(c, &ray);
    }
    c->st.total_energy_received += (double)(col.x + col.y + col.z) * 0.001;   /* ref: Main.cpp:735 */

    if (s->debug_mode == ORC_DEBUG_NONE) {                               /* ref: Main.cpp:738-746 */
        float* a = s->accumulator + 4 * (size_t)pos;
        a[0] += col.x; a[1] += col.y; a[2] += col.z; a[3] += 1.0f;
        float n = (float)s->num_accumulated;
        s->pixels[pos] = vec4_to_uint(a[0] / n, a[1] / n, a[2] / n);
    } else {
        s->pixels[pos] = vec4_to_uint(col.x, col.y, col.z);
    }
}

typedef struct {
    orc_scene* s; int rng_mode; uint32_t seed; uint32_t row_begin, row_end;
    volatile uint32_t* next_job; uint32_t n_jobs, tiles_x;
    orc_stats st;
} worker_t;

static void stats_add(orc_stats* a, const orc_stats* b)
{
    a->traced_rays += b->traced_rays; a->inner_steps += b->inner_steps; a->tri_tests += b->tri_tests;
    a->bvh_depth_sum += b->bvh_depth_sum; a->closest_hits += b->closest_hits;
    a->total_energy_received += b->total_energy_received;
}

static void* pcg_worker(void* arg)
{
    worker_t* w = (worker_t*)arg;
    orc_scene* s = w->s;
    uint32_t pcg_state = 0;
    tctx_t c; memset(&c, 0, sizeof(c));
    c.s = s; c.rng.pcg = 1; c.rng.m = c.rng.u = c.rng.p = &pcg_state;
    float inv_w = 1.0f / (float)s->W, inv_h = 1.0f / (float)s->H;
    for (;;) {
        uint32_t job = __sync_fetch_and_add(w->next_job, 1u);
        if (job >= w->n_jobs) break;
        uint32_t x0 = (job % w->tiles_x) * 16, y0 = w->row_begin + (job / w->tiles_x) * 16;
        for (uint32_t y = y0; y < y0 + 16 && y < w->row_end; ++y)
            for (uint32_t x = x0; x < x0 + 16 && x < s->W; ++x)
                shade_pixel(&c, s, x, y, inv_w, inv_h, ORC_RNG_PIXEL_PCG, w->seed, &pcg_state);
    }
    w->st = c.st;
    return NULL;
}

int orc_render(orc_scene* s, uint32_t W, uint32_t H, uint32_t n_frames, int render_mode, int debug_mode,
               int rng_mode, uint32_t seed, int nthreads, uint32_t row_begin, uint32_t row_end)
{
    if (!s || W == 0 || H == 0) return -1;
    if (s->W != W || s->H != H || !s->accumulator) {
        free(s->accumulator); free(s->pixels);
        s->W = W; s->H = H;
        s->accumulator = (float*)calloc((size_t)W * H * 4, sizeof(float));
        s->pixels = (uint32_t*)calloc((size_t)W * H, sizeof(uint32_t));
        s->num_accumulated = 0;
    }
    s->render_mode = render_mode; s->debug_mode = debug_mode;
    if (row_end > H) row_end = H;
    if (row_begin >= row_end) return -1;

    if (rng_mode == ORC_RNG_REFERENCE_XORSHIFT) {
        /* verbatim job order: needs W%16==0 && H%16==0 (SURVEY A-1), serial, full image */
        if ((W % 16) != 0 || (H % 16) != 0) return -2;
        tctx_t c; memset(&c, 0, sizeof(c));
        c.s = s; c.rng.pcg = 0; c.rng.m = &s->seed_m; c.rng.u = &s->seed_u; c.rng.p = &s->seed_p;
        float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;          /* ref: Main.cpp:700 */
        for (uint32_t f = 0; f < n_frames; ++f) {
            s->num_accumulated++;                                         /* ref: Main.cpp:702 */
            uint32_t num_jobs = (uint32_t)(((size_t)W * H) / 256);        /* ref: Main.cpp:750-751 */
            for (uint32_t job = 0; job < num_jobs; ++job) {
                uint32_t first_x = (job * 16) % W;                        /* ref: Main.cpp:705-706 */
                uint32_t first_y = ((job * 16) / W) * 16;
                for (uint32_t y = first_y; y < first_y + 16; y += 4)      /* ref: Main.cpp:708-711 */
                    for (uint32_t x = first_x; x < first_x + 16; x += 4)
                        for (uint32_t v = 0; v < 4; ++v)
                            for (uint32_t u = 0; u < 4; ++u)
                                shade_pixel(&c, s, x + u, y + v, inv_w, inv_h, rng_mode, seed, NULL);
            }
        }
        stats_add(&s->stats, &c.st);
        return 0;
    }

    if (nthreads < 1) nthreads = 1;
    if (nthreads > 256) nthreads = 256;
    uint32_t tiles_x = (W + 15) / 16, tiles_y = (row_end - row_begin + 15) / 16;
    for (uint32_t f = 0; f < n_frames; ++f) {
        s->num_accumulated++;
        volatile uint32_t next_job = 0;
        worker_t* ws = (worker_t*)calloc((size_t)nthreads, sizeof(worker_t));
        pthread_t* th = (pthread_t*)calloc((size_t)nthreads, sizeof(pthread_t));
        for (int i = 0; i < nthreads; ++i) {
            ws[i].s = s; ws[i].rng_mode = rng_mode; ws[i].seed = seed; ws[i].row_begin = row_begin; ws[i].row_end = row_end;
            ws[i].next_job = &next_job; ws[i].n_jobs = tiles_x * tiles_y; ws[i].tiles_x = tiles_x;
        }
        if (nthreads == 1) {
            pcg_worker(&ws[0]);
        } else {
            for (int i = 0; i < nthreads; ++i) pthread_create(&th[i], NULL, pcg_worker, &ws[i]);
            for (int i = 0; i < nthreads; ++i) pthread_join(th[i], NULL);
        }
        for (int i = 0; i < nthreads; ++i) stats_add(&s->stats, &ws[i].st);
        free(ws); free(th);
    }
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * Scene construction API
 * ---------------------------------------------------------------------------------------------- */
orc_scene* orc_scene_new(void)
{
    orc_scene* s = (orc_scene*)calloc(1, sizeof(orc_scene));
    s->max_ray_depth = 5; s->nee = 1; s->cosine = 1; s->rr = 1;          /* ref: Main.cpp:231-234 */
    s->seed_m = s->seed_u = s->seed_p = 0x12345678u;                      /* ref: Random.h:4 */
    float pos[3] = { 0, 0, 0 }, dir[3] = { 0, 0, -1 };
    orc_set_camera(s, pos, dir, 60.0f, 16.0f / 9.0f);                     /* ref: Main.cpp:152-156 */
    return s;
}

void orc_scene_free(orc_scene* s)
{
    if (!s) return;
    for (uint32_t i = 0; i < s->n_objects; ++i) if (s->objects[i].has_bvh) bvh_free(&s->objects[i].bvh);
    free(s->objects); free(s->materials); free(s->lights); free(s->accumulator); free(s->pixels);
    free(s);
}

static void fill_material(material_t* m, const float albedo[3], float specular, float refractivity,
                          const float absorption[3], float ior, const float emissive[3], float intensity, int is_light)
{
    m->albedo = V3(albedo[0], albedo[1], albedo[2]); m->specular = specular; m->refractivity = refractivity;
    m->absorption = V3(absorption[0], absorption[1], absorption[2]); m->ior = ior;
    m->emissive = V3(emissive[0], emissive[1], emissive[2]); m->intensity = intensity; m->is_light = is_light;
}

int orc_add_material(orc_scene* s, const float albedo[3], float specular, float refractivity,
                     const float absorption[3], float ior, const float emissive[3], float intensity, int is_light)
{
    if (s->n_materials == s->cap_materials) {
        s->cap_materials = s->cap_materials ? s->cap_materials * 2 : 8;
        s->materials = (material_t*)realloc(s->materials, sizeof(material_t) * s->cap_materials);
    }
    fill_material(&s->materials[s->n_materials], albedo, specular, refractivity, absorption, ior, emissive, intensity, is_light);
    return (int)s->n_materials++;
}

int orc_set_material(orc_scene* s, int index, const float albedo[3], float specular, float refractivity,
                     const float absorption[3], float ior, const float emissive[3], float intensity, int is_light)
{
    if (index < 0 || (uint32_t)index >= s->n_materials) return -1;
    fill_material(&s->materials[index], albedo, specular, refractivity, absorption, ior, emissive, intensity, is_light);
    return 0;
}

static object_t* push_object(orc_scene* s)
{
    if (s->n_objects == s->cap_objects) {
        s->cap_objects = s->cap_objects ? s->cap_objects * 2 : 8;
        s->objects = (object_t*)realloc(s->objects, sizeof(object_t) * s->cap_objects);
    }
    object_t* o = &s->objects[s->n_objects++];
    memset(o, 0, sizeof(*o));
    return o;
}

int orc_add_mesh(orc_scene* s, const float* vertices, uint32_t nverts, const uint32_t* indices,
                 uint32_t nindices, uint32_t mat_index, int build_option)
{
    object_t* o = push_object(s);
    o->mat_index = mat_index; o->has_bvh = 1;
    int rc = bvh_build(&o->bvh, vertices, nverts, indices, nindices, build_option);
    if (rc != 0) { bvh_free(&o->bvh); s->n_objects--; return rc; }
    return (int)(s->n_objects - 1);
}

int orc_add_sphere(orc_scene* s, const float center[3], float radius, uint32_t mat_index)
{
    object_t* o = push_object(s);
    o->mat_index = mat_index; o->prim_type = PRIM_SPHERE;
    o->sphere_center = V3(center[0], center[1], center[2]); o->sphere_radius = radius; o->sphere_radius_sq = radius * radius;
    return (int)(s->n_objects - 1);
}

int orc_add_plane(orc_scene* s, const float normal[3], const float point[3], uint32_t mat_index)
{
    object_t* o = push_object(s);
    o->mat_index = mat_index; o->prim_type = PRIM_PLANE;
    o->plane_normal = V3(normal[0], normal[1], normal[2]); o->plane_point = V3(point[0], point[1], point[2]);
    return (int)(s->n_objects - 1);
}

int orc_add_light(orc_scene* s, uint32_t obj_index)
{
    if (obj_index >= s->n_objects) return -1;
    /* only sphere primitives and meshes can be sampled, ref: Main.cpp:371-384 (EXCEPT otherwise) */
    if (!s->objects[obj_index].has_bvh && s->objects[obj_index].prim_type != PRIM_SPHERE) return -2;
    if (s->n_lights == s->cap_lights) {
        s->cap_lights = s->cap_lights ? s->cap_lights * 2 : 8;
        s->lights = (uint32_t*)realloc(s->lights, sizeof(uint32_t) * s->cap_lights);
    }
    s->lights[s->n_lights++] = obj_index;
    return 0;
}

void orc_set_camera(orc_scene* s, const float pos[3], const float view_dir[3], float fov_deg, float aspect)
{
    s->camera.pos = V3(pos[0], pos[1], pos[2]);
    s->camera.view_dir = V3(view_dir[0], view_dir[1], view_dir[2]);
    s->camera.fov = fov_deg * PI_F / 180.0f;                              /* Deg2Rad, ref: MathLib.h:9-12 */
    s->camera.aspect = aspect;
    update_screen_plane(&s->camera);
}

void orc_set_settings(orc_scene* s, int max_ray_depth, int nee, int cosine_weighted, int russian_roulette)
{
    s->max_ray_depth = max_ray_depth; s->nee = nee; s->cosine = cosine_weighted; s->rr = russian_roulette;
}

int orc_rebuild_bvh(orc_scene* s, uint32_t obj_index, int build_option)
{
    if (obj_index >= s->n_objects || !s->objects[obj_index].has_bvh) return -1;
    bvh_rebuild(&s->objects[obj_index].bvh, build_option);
    return 0;
}

int orc_bvh_info_get(const orc_scene* s, uint32_t obj_index, orc_bvh_info* out)
{
    if (obj_index >= s->n_objects || !s->objects[obj_index].has_bvh) return -1;
    const bvh_t* b = &s->objects[obj_index].bvh;
    out->num_triangles = b->n_tris; out->nodes_used = b->current_node; out->max_depth = b->max_depth;
    out->total_area = b->total_area; out->num_leaves = 0; out->max_leaf_size = 0;
    for (uint32_t i = 0; i < b->current_node; ++i) {
        if (b->nodes[i].prim_count > 0) {
            out->num_leaves++;
            if (b->nodes[i].prim_count > out->max_leaf_size) out->max_leaf_size = b->nodes[i].prim_count;
        }
    }
    return 0;
}

int orc_bvh_export(const orc_scene* s, uint32_t obj_index, uint32_t* nodes_words, uint32_t* tri_indices)
{
    if (obj_index >= s->n_objects || !s->objects[obj_index].has_bvh) return -1;
    const bvh_t* b = &s->objects[obj_index].bvh;
    memcpy(nodes_words, b->nodes, sizeof(bvh_node_t) * b->current_node);
    memcpy(tri_indices, b->tri_indices, sizeof(uint32_t) * b->n_tris);
    return 0;
}

void orc_reset_accumulator(orc_scene* s)                    /* ref: Main.cpp:238-243 */
{
    s->num_accumulated = 0;
    if (s->accumulator) memset(s->accumulator, 0, sizeof(float) * 4 * (size_t)s->W * s->H);
    s->stats.total_energy_received = 0.0;
}

const float* orc_accumulator(const orc_scene* s) { return s->accumulator; }
const uint32_t* orc_pixels(const orc_scene* s) { return s->pixels; }
uint32_t orc_num_accumulated(const orc_scene* s) { return s->num_accumulated; }
void orc_get_stats(const orc_scene* s, orc_stats* out) { *out = s->stats; }
void orc_reset_stats(orc_scene* s) { memset(&s->stats, 0, sizeof(s->stats)); }

void orc_intersect_rays(orc_scene* s, const float* origins, const float* dirs, const float* tmax, uint32_t n,
                        float* out_t, uint32_t* out_obj, uint32_t* out_tri, uint32_t* out_depth)
{
    tctx_t c; memset(&c, 0, sizeof(c)); c.s = s;
    for (uint32_t i = 0; i < n; ++i) {
        ray_t r = make_ray(V3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]),
                           V3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]), tmax ? tmax[i] : 1e34f);
        intersect_scene(&c, &r);
        out_t[i] = r.t; out_obj[i] = r.obj_idx; out_tri[i] = r.tri_idx; out_depth[i] = r.bvh_depth;
    }
    stats_add(&s->stats, &c.st);
}

void orc_camera_ray(const orc_scene* s, uint32_t x, uint32_t y, uint32_t W, uint32_t H, float* origin, float* dir)
{
    float inv_w = 1.0f / (float)W, inv_h = 1.0f / (float)H;
    ray_t r = camera_get_ray(&s->camera, (float)x * inv_w, (float)y * inv_h);
    origin[0] = r.origin.x; origin[1] = r.origin.y; origin[2] = r.origin.z;
    dir[0] = r.direction.x; dir[1] = r.direction.y; dir[2] = r.direction.z;
}

/* known-answer entry points */
uint32_t orc_vec4_to_uint(const float v[4]) { return vec4_to_uint(v[0], v[1], v[2]); }
float orc_fresnel(float in, float out, float ior_outside, float ior_inside) { return fresnel(in, out, ior_outside, ior_inside); }
void orc_reflect(const float d[3], const float n[3], float out[3])
{
    v3 r = reflect_dir(V3(d[0], d[1], d[2]), V3(n[0], n[1], n[2])); out[0] = r.x; out[1] = r.y; out[2] = r.z;
}
int orc_intersect_triangle(const float p0[3], const float p1[3], const float p2[3], const float o[3], const float d[3], float* t)
{
    ray_t r = make_ray(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), *t);
    int h = intersect_triangle(V3(p0[0], p0[1], p0[2]), V3(p1[0], p1[1], p1[2]), V3(p2[0], p2[1], p2[2]), &r);
    *t = r.t; return h;
}
int orc_intersect_sphere(const float c[3], float radius, const float o[3], const float d[3], float* t)
{
    ray_t r = make_ray(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), *t);
    int h = intersect_sphere(V3(c[0], c[1], c[2]), radius * radius, &r);
    *t = r.t; return h;
}
float orc_intersect_aabb(const float bmin[3], const float bmax[3], const float o[3], const float d[3], float t)
{
    ray_t r = make_ray(V3(o[0], o[1], o[2]), V3(d[0], d[1], d[2]), t);
    return intersect_aabb_sse(V3(bmin[0], bmin[1], bmin[2]), V3(bmax[0], bmax[1], bmax[2]), &r);
}
