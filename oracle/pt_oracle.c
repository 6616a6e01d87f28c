/*
 * pt_oracle.c — CPU restatement of triSYCL/path_tracer's render() hot path.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/README.md).  PARITY STATUS: the xorshift32
 * generator is pinned against the reference's own header (oracle/_ref/);
 * all float3 arithmetic is "parity unpinned" (the reference needs triSYCL, which
 * is absent; no stand-in headers are written).  float3 operators follow the SYCL
 * meaning with the simplest evaluation order: element-wise + - * /, dot =
 * (x0*y0 + x1*y1) + x2*y2, cross by the textbook formula, length = sqrt(dot).
 * Compile with -ffp-contract=off: the only fused operations are the explicit
 * sycl::fma calls of vec.hpp:12.
 *
 * Every function cites the reference lines it follows (paths relative to
 * /root/reference/include unless noted).
 */
#include "pt_oracle.h"
#include "ptm_portable.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- math mode ------------------------------------------------------------------ */

static int g_portable = 0;
void orc_set_math(int portable) { g_portable = portable ? 1 : 0; }
int orc_get_math(void) { return g_portable; }

static inline float m_sin(float x) { return g_portable ? ptm_sinf(x) : sinf(x); }
static inline float m_cos(float x) { return g_portable ? ptm_cosf(x) : cosf(x); }
static inline float m_log(float x) { return g_portable ? ptm_logf(x) : logf(x); }
static inline float m_pow5(float x) { return g_portable ? ptm_pow5f(x) : powf(x, 5.0f); }
static inline float m_atan2(float y, float x) { return g_portable ? ptm_atan2f(y, x) : atan2f(y, x); }
static inline float m_asin(float x) { return g_portable ? ptm_asinf(x) : asinf(x); }
static inline float m_fmod1(float x) { return g_portable ? ptm_fmod1f(x) : fmodf(x, 1.0f); }

/* ---- float3 (sycl::float3 semantics as pinned above) ------------------------------ */

typedef struct { float x, y, z; } v3;

static inline v3 V(float x, float y, float z) { v3 r = { x, y, z }; return r; }
static inline v3 vld(const float* f) { return V(f[0], f[1], f[2]); }
static inline void vst(float* f, v3 a) { f[0] = a.x; f[1] = a.y; f[2] = a.z; }
static inline v3 vadd(v3 a, v3 b) { return V(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(float s, v3 a) { return V(s * a.x, s * a.y, s * a.z); } /* s*v and v*s */
static inline v3 vdivs(v3 a, float s) { return V(a.x / s, a.y / s, a.z / s); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
  return V(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlength(v3 a) { return sqrtf(vdot(a, a)); }
/* vec.hpp:11-13 — explicit fma */
static inline float length_squared(v3 v) { return fmaf(v.x, v.x, fmaf(v.y, v.y, v.z * v.z)); }
/* vec.hpp:20 */
static inline v3 vneg(v3 u) { return V(-u.x, -u.y, -u.z); }
/* vec.hpp:23 */
static inline v3 unit_vector(v3 v) { return vdivs(v, vlength(v)); }
/* vec.hpp:26 : v - 2*dot(v,n)*n */
static inline v3 reflect(v3 v, v3 n) { return vsub(v, vscale(2.0f * vdot(v, n), n)); }
/* vec.hpp:29-35 */
static inline v3 refract(v3 uv, v3 n, float etai_over_etat) {
  float cos_theta = fminf(-vdot(uv, n), 1.0f);
  v3 r_out_perp = vscale(etai_over_etat, vadd(uv, vscale(cos_theta, n)));
  v3 r_out_parallel = vscale(-sqrtf(fabsf(1.0f - length_squared(r_out_perp))), n);
  return vadd(r_out_perp, r_out_parallel);
}

static const float PT_INF = INFINITY;
static const float PT_PI = 3.1415926535897932385f; /* rtweekend.hpp:22 */

/* ---- per-work-item context (task_context, rtweekend.hpp:99-103) -------------------- */

typedef struct {
  uint32_t rng;
  const PtSceneDesc* sc;
  OrcCounters* c; /* may be NULL */
} ctx_t;

/* xorshift.hpp:64-75 */
static inline uint32_t xs32(uint32_t* s) {
  uint32_t x = *s;
  x ^= x >> 7;
  x ^= x << 1;
  x ^= x >> 9;
  *s = x;
  return x;
}
uint32_t orc_xorshift32(uint32_t* state) { return xs32(state); }

/* rtweekend.hpp:39-42 */
static inline float rng_float(ctx_t* c) {
  const float scale = 1.0f / 4294967296.0f;
  if (c->c) c->c->rng_draws++;
  return (float)xs32(&c->rng) * scale;
}
/* rtweekend.hpp:45-48 */
static inline float rng_float_mm(ctx_t* c, float mn, float mx) { return mn + (mx - mn) * rng_float(c); }
/* rtweekend.hpp:60-67 */
static inline v3 rng_unit_vec(ctx_t* c) {
  float x = rng_float_mm(c, -1.0f, 1.0f);
  float maxy = sqrtf(1.0f - x * x);
  float y = rng_float_mm(c, -maxy, maxy);
  float absz = sqrtf(maxy * maxy - y * y);
  float z = (rng_float(c) > 0.5f) ? absz : -absz;
  return V(x, y, z);
}
/* rtweekend.hpp:70-80 */
static inline v3 rng_in_unit_ball(ctx_t* c) {
  float r = rng_float(c);
  float theta = rng_float_mm(c, 0.0f, 2.0f * PT_PI);
  float phi = rng_float_mm(c, 0.0f, PT_PI);
  float plan_seed = r * m_sin(phi);
  float z = r * m_cos(phi);
  return V(plan_seed * m_cos(theta), plan_seed * m_sin(theta), z);
}
/* rtweekend.hpp:83-88 */
static inline v3 rng_in_unit_disk(ctx_t* c) {
  float x = rng_float_mm(c, -1.0f, 1.0f);
  float maxy = sqrtf(1.0f - x * x);
  float y = rng_float_mm(c, -maxy, maxy);
  return V(x, y, 0.0f);
}

float orc_float_t(uint32_t* state) { ctx_t c = { *state, 0, 0 }; float f = rng_float(&c); *state = c.rng; return f; }
void orc_unit_vec(uint32_t* state, float out[3]) { ctx_t c = { *state, 0, 0 }; vst(out, rng_unit_vec(&c)); *state = c.rng; }
void orc_in_unit_ball(uint32_t* state, float out[3]) { ctx_t c = { *state, 0, 0 }; vst(out, rng_in_unit_ball(&c)); *state = c.rng; }
void orc_in_unit_disk(uint32_t* state, float out[3]) { ctx_t c = { *state, 0, 0 }; vst(out, rng_in_unit_disk(&c)); *state = c.rng; }

/* ---- ray, hit_record (ray.hpp:6-28, hitable.hpp:8-24) -------------------------------- */

typedef struct { v3 orig, dir; float tm; } ray_t;
static inline v3 ray_at(const ray_t* r, float t) { return vadd(r->orig, vscale(t, r->dir)); }

typedef struct {
  float t;
  v3 p, normal;
  int front_face;
  float u, v;
} hit_record;

/* hitable.hpp:20-23 : normal = front_face ? n : vec{} - n */
static inline void set_face_normal(hit_record* rec, const ray_t* r, v3 outward_normal) {
  rec->front_face = vdot(r->dir, outward_normal) < 0;
  rec->normal = rec->front_face ? outward_normal : vsub(V(0.0f, 0.0f, 0.0f), outward_normal);
}

/* ---- sphere (sphere.hpp) ---------------------------------------------------------------- */

/* sphere.hpp:13-24 */
static inline void mercator_coordinates(v3 p, float* u, float* v) {
  float phi = m_atan2(p.z, p.x);
  float theta = m_asin(p.y);
  *u = 1.0f - (phi + PT_PI) / (2.0f * PT_PI);
  *v = (theta + PT_PI / 2.0f) / PT_PI;
}

/* sphere.hpp:51-56 */
static inline v3 sphere_center(const float* f, float time) {
  v3 c0 = vld(f), c1 = vld(f + 3);
  float time0 = f[7], time1 = f[8];
  if (time0 == time1) return c0;
  return vadd(c0, vscale((time - time0) / (time1 - time0), vsub(c1, c0)));
}

/* sphere.hpp:59-106.  want_uv=0 only where the record's u,v are never read
 * (constant_medium boundaries): mercator_coordinates is pure.                 */
static int sphere_hit(ctx_t* c, const float* f, const ray_t* r, float mn, float mx, hit_record* rec, int want_uv) {
  if (c->c) { c->c->sphere_tests++; if (f[7] != f[8]) c->c->sphere_moving++; }
  float radius = f[6];
  v3 oc = vsub(r->orig, sphere_center(f, r->tm));
  float a = vdot(r->dir, r->dir);
  float b = vdot(oc, r->dir);
  float cc = vdot(oc, oc) - radius * radius;
  float discriminant = b * b - a * cc;
  if (discriminant > 0) {
    float temp = (-b - sqrtf(discriminant)) / a;
    if (temp < mx && temp > mn) {
      rec->t = temp;
      rec->p = ray_at(r, rec->t);
      v3 outward_normal = vdivs(vsub(rec->p, sphere_center(f, r->tm)), radius);
      set_face_normal(rec, r, outward_normal);
      if (want_uv) mercator_coordinates(rec->normal, &rec->u, &rec->v);
      if (c->c) c->c->sphere_exit[2]++;
      return 1;
    }
    temp = (-b + sqrtf(discriminant)) / a;
    if (temp < mx && temp > mn) {
      rec->t = temp;
      rec->p = ray_at(r, rec->t);
      v3 outward_normal = vdivs(vsub(rec->p, sphere_center(f, r->tm)), radius);
      set_face_normal(rec, r, outward_normal);
      if (want_uv) mercator_coordinates(rec->normal, &rec->u, &rec->v);
      if (c->c) c->c->sphere_exit[2]++;
      return 1;
    }
    if (c->c) c->c->sphere_exit[1]++;
    return 0;
  }
  if (c->c) c->c->sphere_exit[0]++;
  return 0;
}

/* ---- rectangles (rectangle.hpp:31-49, 69-87, 107-125) ------------------------------------ */
/* axis: 0 = xy_rect (normal +z), 1 = xz_rect (normal +y), 2 = yz_rect (normal +x);
 * a0,a1,b0,b1,k in the constructor's order.                                                  */
static int rect_hit(ctx_t* c, int axis, float a0, float a1, float b0, float b1, float k,
                    const ray_t* r, float mn, float mx, hit_record* rec) {
  if (c->c) c->c->rect_tests++;
  float ok, dk, oa, da, ob, db;
  v3 n;
  if (axis == 0)      { ok = r->orig.z; dk = r->dir.z; oa = r->orig.x; da = r->dir.x; ob = r->orig.y; db = r->dir.y; n = V(0, 0, 1); }
  else if (axis == 1) { ok = r->orig.y; dk = r->dir.y; oa = r->orig.x; da = r->dir.x; ob = r->orig.z; db = r->dir.z; n = V(0, 1, 0); }
  else                { ok = r->orig.x; dk = r->dir.x; oa = r->orig.y; da = r->dir.y; ob = r->orig.z; db = r->dir.z; n = V(1, 0, 0); }
  float t = (k - ok) / dk;
  if (t < mn || t > mx) { if (c->c) c->c->rect_exit[0]++; return 0; }
  float a = oa + t * da;
  float b = ob + t * db;
  if (a < a0 || a > a1 || b < b0 || b > b1) { if (c->c) c->c->rect_exit[1]++; return 0; }
  if (c->c) c->c->rect_exit[2]++;
  rec->u = (a - a0) / (a1 - a0);
  rec->v = (b - b0) / (b1 - b0);
  rec->t = t;
  rec->p = ray_at(r, rec->t);
  set_face_normal(rec, r, n);
  return 1;
}

/* ---- triangle, Moller-Trumbore strategy (triangle.hpp:58-100) ----------------------------- */
static int triangle_hit(ctx_t* c, const float* f, const ray_t* r, float mn, float mx, hit_record* rec) {
  const float epsilon = 0.0000001f;
  v3 v0 = vld(f), v1 = vld(f + 3), v2 = vld(f + 6);
  v3 edge1 = vsub(v1, v0);
  v3 edge2 = vsub(v2, v0);
  v3 h = vcross(r->dir, edge2);
  float a = vdot(edge1, h);
  float a_abs = fabsf(a);
  if (a_abs < epsilon) { if (c->c) c->c->tri_exit[0]++; return 0; }
  int a_pos = a > 0.0f;
  v3 s = vsub(r->orig, v0);
  float u = vdot(s, h);
  int u_pos = u > 0.0f;
  if ((u_pos ^ a_pos) || fabsf(u) > a_abs) { if (c->c) c->c->tri_exit[1]++; return 0; }
  v3 q = vcross(s, edge1);
  float v = vdot(r->dir, q);
  int v_pos = v > 0.0f;
  if ((v_pos ^ a_pos) || (fabsf(u + v) > a_abs)) { if (c->c) c->c->tri_exit[2]++; return 0; }
  float length = vdot(edge2, q) / a;
  if (length < mn || length > mx) { if (c->c) c->c->tri_exit[3]++; return 0; }
  if (c->c) c->c->tri_exit[4]++;
  v3 hit_pt = ray_at(r, length);
  set_face_normal(rec, r, vcross(edge1, edge2)); /* NOT normalised (triangle.hpp:96) */
  rec->t = length;
  rec->p = hit_pt;
  return 1; /* rec->u, rec->v left as they were (stale) */
}

/* ---- triangle, Badouel strategy (triangle.hpp:14-56): the alternative _triangle<> can be instantiated with ---------- */
static int triangle_hit_badouel(ctx_t* c, const float* f, const ray_t* r, float mn, float mx, hit_record* rec) {
  (void)c;
  v3 v0 = vld(f), v1 = vld(f + 3), v2 = vld(f + 6);
  v3 u = vsub(v1, v0);
  v3 v = vsub(v2, v0);
  v3 outward_normal = vcross(u, v);
  v3 w0 = vsub(r->orig, v0);
  float a = -vdot(outward_normal, w0);
  float b = vdot(outward_normal, r->dir);
  if (fabsf(b) < 0.000001f) return 0; /* ray parallel to the plane */
  float length = a / b;
  if (length < 0) return 0;
  else if (length < mn || length > mx) return 0;
  v3 hit_pt = ray_at(r, length);
  float uu = vdot(u, u);
  float uv = vdot(u, v);
  float vv = vdot(v, v);
  v3 w = vsub(hit_pt, v0);
  float wu = vdot(w, u);
  float wv = vdot(w, v);
  float D = uv * uv - uu * vv;
  float s = (uv * wv - vv * wu) / D;
  float t = (uv * wu - uu * wv) / D;
  if (s < 0.0f || s > 1.0f || t < 0.0f || (s + t) > 1.0f) return 0;
  set_face_normal(rec, r, outward_normal);
  rec->t = length;
  rec->p = hit_pt;
  return 1; /* rec->u, rec->v left as they were */
}

/* ---- box (box.hpp:15-50): nearest of six sides in constructor order ------------------------ */
static int box_hit(ctx_t* c, const float* f, const ray_t* r, float mn, float mx, hit_record* rec) {
  float x0 = f[0], y0 = f[1], z0 = f[2], x1 = f[3], y1 = f[4], z1 = f[5];
  hit_record temp_rec;
  int hit_anything = 0;
  float closest_so_far = mx;
  for (int side = 0; side < 6; side++) {
    int h;
    switch (side) {
      case 0: h = rect_hit(c, 0, x0, x1, y0, y1, z1, r, mn, closest_so_far, &temp_rec); break; /* box.hpp:20 */
      case 1: h = rect_hit(c, 0, x0, x1, y0, y1, z0, r, mn, closest_so_far, &temp_rec); break; /* :21 */
      case 2: h = rect_hit(c, 1, x0, x1, z0, z1, y1, r, mn, closest_so_far, &temp_rec); break; /* :22 */
      case 3: h = rect_hit(c, 1, x0, x1, z0, z1, y0, r, mn, closest_so_far, &temp_rec); break; /* :23 */
      case 4: h = rect_hit(c, 2, y0, y1, z0, z1, x1, r, mn, closest_so_far, &temp_rec); break; /* :24 */
      default: h = rect_hit(c, 2, y0, y1, z0, z1, x0, r, mn, closest_so_far, &temp_rec); break; /* :25 */
    }
    if (h) {
      hit_anything = 1;
      closest_so_far = temp_rec.t;
      *rec = temp_rec;
    }
  }
  return hit_anything;
}

/* ---- constant_medium (constant_medium.hpp:28-78) ------------------------------------------- */
static int boundary_hit(ctx_t* c, const PtHittable* h, const ray_t* r, float mn, float mx, hit_record* rec) {
  if (h->boundary_kind == PT_HIT_SPHERE) return sphere_hit(c, h->f, r, mn, mx, rec, 0);
  return box_hit(c, h->f, r, mn, mx, rec);
}

static int medium_hit(ctx_t* c, const PtHittable* h, const ray_t* r, float mn, float mx, hit_record* rec) {
  hit_record rec1, rec2;
  if (!boundary_hit(c, h, r, -PT_INF, PT_INF, &rec1)) return 0;
  if (!boundary_hit(c, h, r, rec1.t + 0.0001f, PT_INF, &rec2)) return 0;
  if (rec1.t < mn) rec1.t = mn;
  if (rec2.t > mx) rec2.t = mx;
  if (rec1.t >= rec2.t) return 0;
  if (rec1.t < 0) rec1.t = 0;
  const float ray_length = vlength(r->dir);
  const float distance_inside_boundary = (rec2.t - rec1.t) * ray_length;
  const float neg_inv_density = h->f[9];
  const float hit_distance = neg_inv_density * m_log(rng_float(c)); /* the in-traversal draw (:65) */
  if (hit_distance > distance_inside_boundary) return 0;
  rec->t = rec1.t + hit_distance / ray_length;
  rec->p = ray_at(r, rec->t);
  rec->normal = V(1, 0, 0);
  rec->front_face = 1;
  return 1; /* u, v untouched */
}

/* ---- hit_world (render.hpp:30-51) ------------------------------------------------------------ */
static int hit_world(ctx_t* c, const ray_t* r, hit_record* rec, int* material, int* hittable) {
  const PtSceneDesc* sc = c->sc;
  hit_record temp_rec;
  memset(&temp_rec, 0, sizeof temp_rec); /* reference leaves it uninitialised; we define 0 */
  int hit_anything = 0;
  float closest_so_far = PT_INF;
  if (c->c) c->c->rays++;
  for (int i = 0; i < sc->n_hittables; i++) {
    const PtHittable* h = &sc->hittables[i];
    int hit;
    if (c->c) c->c->tests[h->kind]++;
    switch (h->kind) {
      case PT_HIT_SPHERE: hit = sphere_hit(c, h->f, r, 0.001f, closest_so_far, &temp_rec, 1); break;
      case PT_HIT_XY_RECT: hit = rect_hit(c, 0, h->f[0], h->f[1], h->f[2], h->f[3], h->f[4], r, 0.001f, closest_so_far, &temp_rec); break;
      case PT_HIT_XZ_RECT: hit = rect_hit(c, 1, h->f[0], h->f[1], h->f[2], h->f[3], h->f[4], r, 0.001f, closest_so_far, &temp_rec); break;
      case PT_HIT_YZ_RECT: hit = rect_hit(c, 2, h->f[0], h->f[1], h->f[2], h->f[3], h->f[4], r, 0.001f, closest_so_far, &temp_rec); break;
      case PT_HIT_TRIANGLE: /* _triangle<IntersectionStrategy>::hit triangle.hpp:113-117 */
        hit = h->strategy == PT_TRI_BADOUEL ? triangle_hit_badouel(c, h->f, r, 0.001f, closest_so_far, &temp_rec)
                                            : triangle_hit(c, h->f, r, 0.001f, closest_so_far, &temp_rec);
        break;
      case PT_HIT_BOX: hit = box_hit(c, h->f, r, 0.001f, closest_so_far, &temp_rec); break;
      default: hit = medium_hit(c, h, r, 0.001f, closest_so_far, &temp_rec); break;
    }
    if (hit) {
      if (c->c) c->c->accepts[h->kind]++;
      hit_anything = 1;
      closest_so_far = temp_rec.t;
      *rec = temp_rec;
      *material = h->material;
      *hittable = i;
    }
  }
  return hit_anything;
}

/* ---- textures (texture.hpp:25, 42-49, 135-151) ------------------------------------------------ */
static inline uint32_t texel_index(float f, uint32_t maxv) {
  /* (size_t)f of the reference for 0 <= f <= maxv; NaN/negative (UB there) -> 0, overflow -> maxv */
  if (!(f > 0.0f)) return 0;
  if (f >= (float)maxv) return maxv;
  return (uint32_t)f;
}

static v3 texture_value(ctx_t* c, int tex, const hit_record* rec) {
  const PtTexture* t = &c->sc->textures[tex];
  if (c->c) c->c->tex_evals[t->kind]++;
  if (t->kind == PT_TEX_SOLID) return vld(t->color0);
  if (t->kind == PT_TEX_CHECKER) {
    float sines = m_sin(10.0f * rec->p.x) * m_sin(10.0f * rec->p.y) * m_sin(10.0f * rec->p.z);
    if (sines < 0) return vld(t->color0); /* odd */
    return vld(t->color1);                /* even */
  }
  /* image_texture::value texture.hpp:135-151 */
  uint32_t i = texel_index(m_fmod1(rec->u * t->freq) * (float)(t->width - 1), t->width - 1);
  uint32_t j = texel_index((1.0f - m_fmod1(rec->v * t->freq)) * (float)(t->height - 1), t->height - 1);
  uint64_t pix_idx = (uint64_t)j * t->width + i + t->offset;
  const float scale = 1.0f / 255;
  const uint8_t* d = c->sc->atlas;
  return V((float)d[pix_idx * 3] * scale, (float)d[pix_idx * 3 + 1] * scale, (float)d[pix_idx * 3 + 2] * scale);
}

/* ---- materials (material.hpp) ------------------------------------------------------------------ */

/* material.hpp:62-66 */
static inline float reflectance(float cosine, float ref_idx) {
  float r0 = (1.0f - ref_idx) / (1.0f + ref_idx);
  r0 *= r0;
  return r0 + (1.0f - r0) * m_pow5(1.0f - cosine);
}

static v3 material_emitted(ctx_t* c, int mat, const hit_record* rec) {
  const PtMaterial* m = &c->sc->materials[mat];
  if (m->kind == PT_MAT_LIGHTSOURCE) return texture_value(c, m->texture, rec); /* material.hpp:106-108 */
  return V(0, 0, 0);
}

static int material_scatter(ctx_t* c, int mat, const ray_t* r_in, const hit_record* rec, v3* attenuation, ray_t* scattered) {
  const PtMaterial* m = &c->sc->materials[mat];
  if (c->c) c->c->scatters[m->kind]++;
  switch (m->kind) {
    case PT_MAT_LAMBERTIAN: { /* material.hpp:18-28 */
      v3 scatter_direction = vadd(rec->normal, rng_unit_vec(c));
      scattered->orig = rec->p; scattered->dir = scatter_direction; scattered->tm = r_in->tm;
      *attenuation = vmul(*attenuation, texture_value(c, m->texture, rec));
      return 1;
    }
    case PT_MAT_METAL: { /* material.hpp:39-48 */
      v3 reflected = reflect(unit_vector(r_in->dir), rec->normal);
      v3 ball = rng_in_unit_ball(c);
      scattered->orig = rec->p; scattered->dir = vadd(reflected, vscale(m->param, ball)); scattered->tm = r_in->tm;
      *attenuation = vmul(*attenuation, vld(m->color));
      return vdot(scattered->dir, rec->normal) > 0;
    }
    case PT_MAT_DIELECTRIC: { /* material.hpp:68-88 */
      *attenuation = vmul(*attenuation, vld(m->color));
      float ref_idx = m->param;
      float refraction_ratio = rec->front_face ? (1.0f / ref_idx) : ref_idx;
      v3 unit_direction = unit_vector(r_in->dir);
      float cos_theta = fminf(-vdot(unit_direction, rec->normal), 1.0f);
      float sin_theta = sqrtf(1.0f - cos_theta * cos_theta);
      int cannot_refract = refraction_ratio * sin_theta > 1.0f;
      v3 direction;
      if (cannot_refract || reflectance(cos_theta, refraction_ratio) > rng_float(c))
        direction = reflect(unit_direction, rec->normal);
      else
        direction = refract(unit_direction, rec->normal, refraction_ratio);
      scattered->orig = rec->p; scattered->dir = direction; scattered->tm = r_in->tm;
      return 1;
    }
    case PT_MAT_LIGHTSOURCE: /* material.hpp:104 */
      return 0;
    default: { /* isotropic, material.hpp:119-126 */
      v3 ball = rng_in_unit_ball(c);
      scattered->orig = rec->p; scattered->dir = ball; scattered->tm = r_in->tm;
      *attenuation = vmul(*attenuation, texture_value(c, m->texture, rec));
      return 1;
    }
  }
}

/* ---- sky (render.hpp:83-87) ----------------------------------------------------------------------- */
static inline v3 sky_color(const ray_t* r, v3 cur_attenuation) {
  v3 unit_direction = unit_vector(r->dir);
  float hit_pt = 0.5f * (unit_direction.y + 1.0f);
  v3 col = vadd(vscale(1.0f - hit_pt, V(1.0f, 1.0f, 1.0f)), vscale(hit_pt, V(0.5f, 0.7f, 1.0f)));
  return vmul(cur_attenuation, col);
}

/* ---- get_color (render.hpp:29-92) ------------------------------------------------------------------- */
static v3 get_color(ctx_t* c, const ray_t* r, int depth) {
  ray_t cur_ray = *r;
  v3 cur_attenuation = V(1.0f, 1.0f, 1.0f);
  for (int i = 0; i < depth; i++) {
    hit_record rec;
    memset(&rec, 0, sizeof rec);
    int material = -1, hittable = -1;
    if (hit_world(c, &cur_ray, &rec, &material, &hittable)) {
      v3 emitted = material_emitted(c, material, &rec);
      ray_t scattered;
      if (material_scatter(c, material, &cur_ray, &rec, &cur_attenuation, &scattered)) {
        cur_ray = scattered;
      } else {
        if (c->c) c->c->end_emit++;
        return emitted; /* NOT multiplied by the attenuation (render.hpp:73) */
      }
    } else {
      if (c->c) c->c->end_sky++;
      return sky_color(&cur_ray, cur_attenuation);
    }
  }
  if (c->c) c->c->end_depth++;
  return V(0.0f, 0.0f, 0.0f);
}

/* ---- camera (camera.hpp) ------------------------------------------------------------------------------ */

/* camera.hpp:67-87 */
void orc_camera_init(PtCamera* cam, const float look_from[3], const float look_at[3], const float vup_[3],
                     float degree_vfov, float aspect_ratio, float aperture, float focus_dist,
                     float time0, float time1) {
  v3 origin = vld(look_from);
  float theta = degree_vfov * PT_PI / 180.0f; /* rtweekend.hpp:31 */
  float h = tanf(theta / 2.0f);
  float viewport_height = 2.0f * h;
  float viewport_width = aspect_ratio * viewport_height;
  v3 w = unit_vector(vsub(vld(look_from), vld(look_at)));
  v3 u = unit_vector(vcross(vld(vup_), w));
  v3 v = vcross(w, u);
  v3 horizontal = vscale(focus_dist * viewport_width, u);
  v3 vertical = vscale(focus_dist * viewport_height, v);
  v3 llc = vsub(vsub(vsub(origin, vdivs(horizontal, 2.0f)), vdivs(vertical, 2.0f)), vscale(focus_dist, w));
  vst(cam->origin, origin);
  vst(cam->lower_left_corner, llc);
  vst(cam->horizontal, horizontal);
  vst(cam->vertical, vertical);
  vst(cam->u, u); vst(cam->v, v); vst(cam->w, w);
  cam->lens_radius = aperture / 2.0f;
  cam->time0 = time0;
  cam->time1 = time1;
}

/* camera.hpp:93-100 */
static inline ray_t camera_get_ray(const PtCamera* cam, float s, float t, ctx_t* c) {
  v3 rd = vscale(cam->lens_radius, rng_in_unit_disk(c));
  v3 offset = vadd(vscale(rd.x, vld(cam->u)), vscale(rd.y, vld(cam->v)));
  ray_t r;
  v3 origin = vld(cam->origin);
  r.orig = vadd(origin, offset);
  r.dir = vsub(vsub(vadd(vadd(vld(cam->lower_left_corner), vscale(s, vld(cam->horizontal))),
                         vscale(t, vld(cam->vertical))), origin), offset);
  r.tm = rng_float_mm(c, cam->time0, cam->time1);
  return r;
}

/* first ray of a sample: render.hpp:96-99 */
static inline ray_t sample_ray(const PtCamera* cam, int x, int y, int width, int height, ctx_t* c) {
  const float u = ((float)x + rng_float(c)) / (float)width;
  const float v = ((float)y + rng_float(c)) / (float)height;
  return camera_get_ray(cam, u, v, c);
}

/* ---- render_pixel (render.hpp:25-106) + executor's seeding (render.hpp:124-136) ------------------------ */
static void render_pixel(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, int x, int y,
                         float out[3], OrcCounters* ctr) {
  ctx_t c;
  c.rng = (uint32_t)((uint64_t)y * (uint64_t)p->width + (uint64_t)x); /* std::hash<size_t> is the identity; LocalPseudoRNG takes uint32 */
  c.sc = sc;
  c.c = ctr;
  v3 final_color = V(0.0f, 0.0f, 0.0f);
  for (int i = 0; i < p->samples; i++) {
    ray_t r = sample_ray(cam, x, y, p->width, p->height, &c);
    final_color = vadd(final_color, get_color(&c, &r, p->depth));
    if (ctr) ctr->samples++;
  }
  final_color = vdivs(final_color, (float)p->samples);
  vst(out, final_color);
}

/* PT_FLAG_FAST_RNG (include/pt_render.h): NOT the reference's image — the opt-in decorrelated mode, restated here so that
 * the GPU's fast mode has a bit-exact checker of its own.  Chunks of PT_FAST_CHUNK_SPP samples, each with its own stream;
 * chunk sums (sequential float adds from 0) are added in chunk order, then one division by the sample count.            */
/* the oracle's own statement of pt_fast_seed (include/pt_render.h): nothing of the product is linked here */
static inline uint32_t orc_fast_seed(uint32_t pixel, uint32_t chunk) {
  uint32_t h = pixel * 0x9E3779B1u + chunk * 0x85EBCA77u + 0x165667B1u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h ? h : 1u;
}

static void render_pixel_fast(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, int x, int y,
                              float out[3], OrcCounters* ctr) {
  ctx_t c;
  c.sc = sc;
  c.c = ctr;
  const uint32_t id = (uint32_t)((uint64_t)y * (uint64_t)p->width + (uint64_t)x);
  v3 total = V(0.0f, 0.0f, 0.0f);
  for (int chunk = 0, s0 = 0; s0 < p->samples; chunk++, s0 += PT_FAST_CHUNK_SPP) {
    const int n = p->samples - s0 < PT_FAST_CHUNK_SPP ? p->samples - s0 : PT_FAST_CHUNK_SPP;
    c.rng = orc_fast_seed(id, (uint32_t)chunk);
    v3 sum = V(0.0f, 0.0f, 0.0f);
    for (int i = 0; i < n; i++) {
      ray_t r = sample_ray(cam, x, y, p->width, p->height, &c);
      sum = vadd(sum, get_color(&c, &r, p->depth));
      if (ctr) ctr->samples++;
    }
    total = vadd(total, sum);
  }
  vst(out, vdivs(total, (float)p->samples));
}

/* render.hpp:113-122 (USE_SINGLE_TASK): one LocalPseudoRNG, default-seeded (xorshift.hpp:18), shared by every pixel; the
 * loops run x outer, y inner; render_pixel is the same function, it just receives the shared context.                  */
static void render_single_stream(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, float* fb, OrcCounters* ctr) {
  ctx_t c;
  c.rng = 2463534242u;
  c.sc = sc;
  c.c = ctr;
  for (int x = 0; x != p->width; ++x)
    for (int y = 0; y != p->height; ++y) {
      v3 final_color = V(0.0f, 0.0f, 0.0f);
      for (int i = 0; i < p->samples; i++) {
        ray_t r = sample_ray(cam, x, y, p->width, p->height, &c);
        final_color = vadd(final_color, get_color(&c, &r, p->depth));
        if (ctr) ctr->samples++;
      }
      vst(fb + ((int64_t)y * p->width + x) * 3, vdivs(final_color, (float)p->samples));
    }
}

static void render_pixel_any(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, int x, int y,
                             float out[3], OrcCounters* ctr) {
  if (p->flags & PT_FLAG_FAST_RNG) render_pixel_fast(sc, cam, p, x, y, out, ctr);
  else render_pixel(sc, cam, p, x, y, out, ctr);
}

static int validate(const PtSceneDesc* sc) {
  if (!sc || sc->n_hittables < 0 || sc->n_materials < 0 || sc->n_textures < 0) return PT_ERR_INVALID_ARG;
  for (int i = 0; i < sc->n_textures; i++) {
    const PtTexture* t = &sc->textures[i];
    if (t->kind < 0 || t->kind > PT_TEX_IMAGE) return PT_ERR_BAD_SCENE;
    if (t->kind == PT_TEX_IMAGE) {
      if (t->width < 1 || t->height < 1) return PT_ERR_BAD_SCENE;
      if (((uint64_t)t->offset + (uint64_t)t->width * t->height) * 3 > sc->atlas_bytes) return PT_ERR_BAD_SCENE;
    }
  }
  for (int i = 0; i < sc->n_materials; i++) {
    const PtMaterial* m = &sc->materials[i];
    if (m->kind < 0 || m->kind > PT_MAT_ISOTROPIC) return PT_ERR_BAD_SCENE;
    int needs_tex = m->kind == PT_MAT_LAMBERTIAN || m->kind == PT_MAT_LIGHTSOURCE || m->kind == PT_MAT_ISOTROPIC;
    if (needs_tex && (m->texture < 0 || m->texture >= sc->n_textures)) return PT_ERR_BAD_SCENE;
  }
  for (int i = 0; i < sc->n_hittables; i++) {
    const PtHittable* h = &sc->hittables[i];
    if (h->kind < 0 || h->kind >= PT_HIT_KIND_COUNT) return PT_ERR_BAD_SCENE;
    if (h->material < 0 || h->material >= sc->n_materials) return PT_ERR_BAD_SCENE;
    if (h->kind == PT_HIT_CONSTANT_MEDIUM && h->boundary_kind != PT_HIT_SPHERE && h->boundary_kind != PT_HIT_BOX)
      return PT_ERR_BAD_SCENE;
    if (h->kind == PT_HIT_TRIANGLE && h->strategy != PT_TRI_MOLLER_TRUMBORE && h->strategy != PT_TRI_BADOUEL) return PT_ERR_BAD_SCENE;
  }
  return PT_OK;
}

static void add_counters(OrcCounters* dst, const OrcCounters* src) {
  uint64_t* d = (uint64_t*)dst;
  const uint64_t* s = (const uint64_t*)src;
  for (size_t i = 0; i < sizeof(OrcCounters) / sizeof(uint64_t); i++) d[i] += s[i];
}

void orc_set_threads(int n) { /* bench.py's cpu_baseline: the 1-thread rate beside the all-threads one */
#ifdef _OPENMP
  if (n > 0) omp_set_num_threads(n);
#else
  (void)n;
#endif
}
int orc_max_threads(void) {
#ifdef _OPENMP
  return omp_get_max_threads();
#else
  return 1;
#endif
}

/* Where pixel (x,y) lands in the caller's buffer, or -1 if it is not this shard's. */
static inline int64_t fb_index(const PtRenderParams* p, int x, int y) {
  if (p->shard_count <= 1) return ((int64_t)y * p->width + x) * 3;
  int tiles_x = (p->width + PT_TILE - 1) / PT_TILE;
  int g = (y / PT_TILE) * tiles_x + (x / PT_TILE);
  if (g % p->shard_count != p->shard_index) return -1;
  int64_t l = g / p->shard_count;
  return (l * PT_TILE_PIXELS + (y % PT_TILE) * PT_TILE + (x % PT_TILE)) * 3;
}

static int render_rows(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, int y0, int y1,
                       float* fb, int64_t row_base, OrcCounters* counters) {
  int rc = validate(sc);
  if (rc) return rc;
  if (!cam || !p || !fb || p->width <= 0 || p->height <= 0 || p->samples <= 0 || p->depth < 0 ||
      p->shard_count < 1 || p->shard_index < 0 || p->shard_index >= p->shard_count)
    return PT_ERR_INVALID_ARG;
  if (counters) memset(counters, 0, sizeof *counters);
  if (p->flags & PT_FLAG_SINGLE_STREAM) { /* the reference's single-task executor: one stream, inherently sequential */
    if (p->shard_count != 1 || y0 != 0 || y1 != p->height) return PT_ERR_INVALID_ARG;
    render_single_stream(sc, cam, p, fb, counters);
    return PT_OK;
  }
#pragma omp parallel
  {
    OrcCounters local;
    memset(&local, 0, sizeof local);
#pragma omp for schedule(dynamic, 1)
    for (int y = y0; y < y1; y++) {
      for (int x = 0; x < p->width; x++) {
        int64_t idx = fb_index(p, x, y);
        if (idx < 0) continue;
        render_pixel_any(sc, cam, p, x, y, fb + idx - row_base, counters ? &local : NULL);
      }
    }
    if (counters) {
#pragma omp critical
      add_counters(counters, &local);
    }
  }
  return PT_OK;
}

int orc_render(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, float* fb, OrcCounters* counters) {
  if (!p) return PT_ERR_INVALID_ARG;
  if (p->shard_count > 1 && fb) { /* padding pixels of edge tiles and the padded last tile are 0 */
    int tiles_x = (p->width + PT_TILE - 1) / PT_TILE, tiles_y = (p->height + PT_TILE - 1) / PT_TILE;
    int64_t per = ((int64_t)tiles_x * tiles_y + p->shard_count - 1) / p->shard_count;
    memset(fb, 0, (size_t)per * PT_TILE_PIXELS * 3 * sizeof(float));
  }
  return render_rows(sc, cam, p, 0, p->height, fb, 0, counters);
}

int orc_render_rows(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, int32_t y0, int32_t y1,
                    float* fb_rows, OrcCounters* counters) {
  if (!p || p->shard_count != 1 || y0 < 0 || y1 > p->height || y0 > y1) return PT_ERR_INVALID_ARG;
  return render_rows(sc, cam, p, y0, y1, fb_rows, (int64_t)y0 * p->width * 3, counters);
}

/* Arbitrary pixels of the frame (full spp each): lets a test check a 1080p x 1024 spp GPU frame at
 * sampled positions in seconds. xy = [n][2], out = [n][3].                                            */
int orc_render_pixels(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, const int32_t* xy,
                      int32_t n, float* out) {
  int rc = validate(sc);
  if (rc) return rc;
  if (!cam || !p || !xy || !out || p->width <= 0 || p->height <= 0 || p->samples <= 0 || p->depth < 0)
    return PT_ERR_INVALID_ARG;
#pragma omp parallel for schedule(dynamic, 4)
  for (int32_t k = 0; k < n; k++) render_pixel_any(sc, cam, p, xy[2 * k], xy[2 * k + 1], out + 3 * (int64_t)k, NULL);
  return PT_OK;
}

/* As orc_render_pixels, plus the number of rays (hit_world calls, render.hpp:60) each pixel traced: the length of
 * that pixel's sequential chain, which bounds any schedule of the frame.                 */
int orc_render_pixels_rays(const PtSceneDesc* sc, const PtCamera* cam, const PtRenderParams* p, const int32_t* xy,
                           int32_t n, float* out, uint64_t* rays) {
  int rc = validate(sc);
  if (rc) return rc;
  if (!cam || !p || !xy || !out || !rays || p->width <= 0 || p->height <= 0 || p->samples <= 0 || p->depth < 0)
    return PT_ERR_INVALID_ARG;
#pragma omp parallel for schedule(dynamic, 4)
  for (int32_t k = 0; k < n; k++) {
    OrcCounters c;
    memset(&c, 0, sizeof c);
    render_pixel_any(sc, cam, p, xy[2 * k], xy[2 * k + 1], out + 3 * (int64_t)k, &c);
    rays[k] = c.rays;
  }
  return PT_OK;
}

/* ---- function-level probes ---------------------------------------------------------------------------- */

int orc_bounce(const PtSceneDesc* sc, const PtBounceIn* in, PtBounceOut* out, int32_t n, int32_t depth_unused) {
  (void)depth_unused;
  int rc = validate(sc);
  if (rc) return rc;
  for (int32_t k = 0; k < n; k++) {
    ctx_t c = { in[k].rng_state, sc, NULL };
    ray_t r;
    r.orig = vld(in[k].origin); r.dir = vld(in[k].dir); r.tm = in[k].time;
    v3 att = vld(in[k].attenuation);
    PtBounceOut* o = &out[k];
    memset(o, 0, sizeof *o);
    hit_record rec;
    memset(&rec, 0, sizeof rec);
    int material = -1, hittable = -1;
    if (hit_world(&c, &r, &rec, &material, &hittable)) {
      o->hittable = hittable; o->material = material; o->front_face = rec.front_face;
      o->t = rec.t; vst(o->p, rec.p); vst(o->normal, rec.normal); o->u = rec.u; o->v = rec.v;
      v3 emitted = material_emitted(&c, material, &rec);
      ray_t sc_ray;
      if (material_scatter(&c, material, &r, &rec, &att, &sc_ray)) {
        o->status = PT_BOUNCE_SCATTERED;
        vst(o->color, att);
        vst(o->sc_origin, sc_ray.orig); vst(o->sc_dir, sc_ray.dir); o->sc_time = sc_ray.tm;
      } else {
        o->status = PT_BOUNCE_ABSORBED;
        vst(o->color, emitted);
      }
    } else {
      o->status = PT_BOUNCE_MISS;
      o->hittable = -1; o->material = -1;
      vst(o->color, sky_color(&r, att));
    }
    o->rng_state = c.rng;
  }
  return PT_OK;
}

int orc_camera_rays(const PtCamera* cam, int32_t width, int32_t height, const int32_t* xy, const uint32_t* rng_in,
                    PtCameraRay* out, int32_t n) {
  for (int32_t k = 0; k < n; k++) {
    ctx_t c = { rng_in[k], NULL, NULL };
    ray_t r = sample_ray(cam, xy[2 * k], xy[2 * k + 1], width, height, &c);
    vst(out[k].origin, r.orig); vst(out[k].dir, r.dir); out[k].time = r.tm; out[k].rng_state = c.rng;
  }
  return PT_OK;
}

int orc_math(int32_t op, const float* a, const float* b, float* out, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    switch (op) {
      case 0: out[i] = m_sin(a[i]); break;
      case 1: out[i] = m_cos(a[i]); break;
      case 2: out[i] = m_log(a[i]); break;
      case 3: out[i] = m_pow5(a[i]); break;
      case 4: out[i] = m_atan2(a[i], b[i]); break;
      case 5: out[i] = m_asin(a[i]); break;
      case 6: out[i] = m_fmod1(a[i]); break;
      case 7: out[i] = sqrtf(a[i]); break;
      case 8: out[i] = a[i] / b[i]; break;
      default: return PT_ERR_INVALID_ARG;
    }
  }
  return PT_OK;
}

/* main.cpp:33-59 : sqrt, clamp [0,0.999], *256, truncate; rows flipped.  std::clamp(NaN) returns NaN;
 * int(NaN) is UB there — defined here as 0.                                                             */
void orc_tonemap_rgb8(const float* fb, int32_t width, int32_t height, uint8_t* rgb8) {
  int64_t index = 0;
  for (int j = height - 1; j >= 0; --j) {
    for (int i = 0; i < width; ++i) {
      for (int ch = 0; ch < 3; ch++) {
        float s = sqrtf(fb[((int64_t)j * width + i) * 3 + ch]);
        float cl = (s < 0.0f) ? 0.0f : (0.999f < s) ? 0.999f : s; /* std::clamp(v, lo, hi) */
        float sc = 256.0f * cl;
        int v = (sc == sc) ? (int)sc : 0;
        rgb8[index++] = (uint8_t)v;
      }
    }
  }
}
