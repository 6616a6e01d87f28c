/*
 * pt_render.h — C-ABI drop-in boundary of the MI355X-native render() hot path.
 *
 * What it replaces (all citations are into the reference tree, triSYCL/path_tracer):
 *
 *   render<width,height,samples>(sycl::queue&, sycl::buffer<color,2>&,
 *                                std::vector<hittable_t>&, camera&)      include/render.hpp:141-160
 *
 * The reference has no FFI/plugin layer; its boundary is that one C++ function
 * template plus a hidden third input, the process-global image-texture atlas
 * (include/texture.hpp:71,126-131,157).  This header states the same boundary as a
 * plain C ABI (pointers + sizes, no C++/torch types) so any host — the C++20
 * facade in path_tracer_amd/include/pt/, the Python ctypes mirror in
 * path_tracer_amd/, or the reference's own render.hpp through the adapter of
 * INTEGRATION.md (which needs what the reference does not expose today: read access to
 * camera's ten private fields, camera.hpp:23-51, and to image_texture's fields) — can
 * bind it.  The facade's render<W,H,S>(frame_buf, hittables, cam) keeps the reference's
 * argument order but has no sycl::queue& / sycl::buffer& (there is no SYCL here).
 *
 * The scene crosses the boundary as three small tables (hittables, materials,
 * textures) + the RGB8 atlas; tags keep the reference's std::variant index order
 * (render.hpp:22-23, material.hpp:133-135, texture.hpp:154, rectangle.hpp:130,
 * constant_medium.hpp:10) so dumps are debuggable against the reference.
 * pt_scene_create() flattens those tables into per-kind 16-byte-aligned record
 * arrays + an order-preserving run table in HBM (see DESIGN.md "Data layout").
 *
 * All floating point is IEEE binary32, no contraction; RNG is xorshift32
 * (include/xorshift.hpp:72-74) seeded with the pixel's linear id (render.hpp:130-132).
 */
#ifndef PT_RENDER_H
#define PT_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2 (round 6): PtTuning's word 9 is tri_binned (was tri_Mg, dead since round 5), tri_cache appended; pt_scene_device_bytes, pt_build_id.
 * The table structs (PtHittable / PtMaterial / PtTexture / PtCamera / PtRenderParams) are those of version 1: scene fixtures written
 * under version 1 load unchanged (path_tracer_amd/scene_io.py).                                                                    */
#define PT_ABI_VERSION 2

/* ---- tags: same numbering as the reference's variant alternatives ---------- */

/* hittable_t = variant<sphere, xy_rect, triangle, box, constant_medium>  render.hpp:22-23 */
enum {
  PT_HIT_SPHERE = 0,          /* sphere.hpp:26-117 (static + moving)             */
  PT_HIT_XY_RECT = 1,         /* rectangle.hpp:16-52                              */
  PT_HIT_TRIANGLE = 2,        /* triangle.hpp:58-122 (Moller-Trumbore strategy)   */
  PT_HIT_BOX = 3,             /* box.hpp:10-56                                    */
  PT_HIT_CONSTANT_MEDIUM = 4, /* constant_medium.hpp:16-83                        */
  /* Extension: classes exist (rectangle.hpp:54,92) but are not hittable_t
   * alternatives in the reference; offered as top-level hittables here.        */
  PT_HIT_XZ_RECT = 5,
  PT_HIT_YZ_RECT = 6,
  PT_HIT_KIND_COUNT = 7
};

/* _triangle<IntersectionStrategy> (triangle.hpp:102-103): moller_trumbore_triangle_intersec is the default and the only one
 * main.cpp instantiates; badouel_ray_triangle_intersec (triangle.hpp:14-56) is the alternative the header ships.        */
enum { PT_TRI_MOLLER_TRUMBORE = 0, PT_TRI_BADOUEL = 1 };

/* material_t = variant<lambertian, metal, dielectric, lightsource, isotropic>  material.hpp:133-135 */
enum {
  PT_MAT_LAMBERTIAN = 0,
  PT_MAT_METAL = 1,
  PT_MAT_DIELECTRIC = 2,
  PT_MAT_LIGHTSOURCE = 3,
  PT_MAT_ISOTROPIC = 4
};

/* texture_t = variant<checker_texture, solid_texture, image_texture>  texture.hpp:154 */
enum { PT_TEX_CHECKER = 0, PT_TEX_SOLID = 1, PT_TEX_IMAGE = 2 };

/* ---- scene description tables (host memory, caller-owned, read-only) ------- */

/* One hittable, 64 bytes.  f[] by kind:
 *  SPHERE           f[0..2]=center0 f[3..5]=center1 f[6]=radius f[7]=time0 f[8]=time1
 *                   (static sphere: center1=center0, time0=time1=0  sphere.hpp:30-36)
 *  XY_RECT          f[0]=x0 f[1]=x1 f[2]=y0 f[3]=y1 f[4]=k          rectangle.hpp:21
 *  XZ_RECT          f[0]=x0 f[1]=x1 f[2]=z0 f[3]=z1 f[4]=k          rectangle.hpp:59
 *  YZ_RECT          f[0]=y0 f[1]=y1 f[2]=z0 f[3]=z1 f[4]=k          rectangle.hpp:97
 *  TRIANGLE         f[0..2]=v0 f[3..5]=v1 f[6..8]=v2                triangle.hpp:107
 *  BOX              f[0..2]=p0 (min) f[3..5]=p1 (max)               box.hpp:15
 *  CONSTANT_MEDIUM  f[0..8]=boundary in SPHERE or BOX layout (boundary_kind),
 *                   f[9]=neg_inv_density (= -1/d, constant_medium.hpp:20);
 *                   material = the isotropic phase function         constant_medium.hpp:21
 */
typedef struct PtHittable {
  int32_t kind;
  int32_t material;      /* index into PtSceneDesc.materials */
  int32_t boundary_kind; /* CONSTANT_MEDIUM only: PT_HIT_SPHERE or PT_HIT_BOX */
  int32_t strategy;      /* TRIANGLE only: the template argument of _triangle<> (triangle.hpp:102-103): PT_TRI_MOLLER_TRUMBORE
                          * (= `triangle`, what main.cpp builds) or PT_TRI_BADOUEL (triangle.hpp:14-56); 0 elsewhere      */
  float f[12];
} PtHittable;

/* One material, 32 bytes.
 *  LAMBERTIAN   texture = albedo                              material.hpp:11-31
 *  METAL        color = albedo, param = fuzz (already clamped to [0,1], material.hpp:37)
 *  DIELECTRIC   color = albedo, param = ref_idx               material.hpp:55-95
 *  LIGHTSOURCE  texture = emit                                material.hpp:97-111
 *  ISOTROPIC    texture = albedo                              material.hpp:113-131
 */
typedef struct PtMaterial {
  int32_t kind;
  int32_t texture; /* index into PtSceneDesc.textures, or -1 */
  float color[3];
  float param;
  int32_t reserved[2];
} PtMaterial;

/* One texture, 48 bytes.
 *  SOLID    color0                                            texture.hpp:18-29
 *  CHECKER  color0 = odd (sines < 0), color1 = even           texture.hpp:32-52
 *  IMAGE    width,height,offset (in texels into the atlas),freq  texture.hpp:67-152
 */
typedef struct PtTexture {
  int32_t kind;
  float color0[3];
  float color1[3];
  uint32_t width, height;
  uint32_t offset;
  float freq;
  int32_t reserved;
} PtTexture;

typedef struct PtSceneDesc {
  const PtHittable* hittables; /* list order == traversal order (render.hpp:37) */
  int32_t n_hittables;
  const PtMaterial* materials;
  int32_t n_materials;
  const PtTexture* textures;
  int32_t n_textures;
  int32_t reserved;
  /* RGB8 atlas, rows top-down; texel 0 is the {0,0,1} load-failure fallback
   * (texture.hpp:157).  May be NULL/0 when no image texture is used.          */
  const uint8_t* atlas;
  uint64_t atlas_bytes;
} PtSceneDesc;

/* camera, 96 bytes: the private fields of camera.hpp:23-51 in declaration order.
 * pt_camera_init() is the 9-argument constructor camera.hpp:67-87.             */
typedef struct PtCamera {
  float origin[3];
  float lower_left_corner[3];
  float horizontal[3];
  float vertical[3];
  float u[3], v[3], w[3];
  float lens_radius;
  float time0, time1;
} PtCamera;

/* Pixels are grouped into 8x8 tiles (one 64-lane wavefront per tile).  Tile g =
 * ty*tiles_x + tx belongs to shard g % shard_count, local index g / shard_count. */
#define PT_TILE 8
#define PT_TILE_PIXELS 64

enum {
  PT_FLAG_NONE = 0,
  PT_FLAG_NO_LDS = 1u << 0,       /* A/B switch: fetch primitives with scalar loads instead of LDS */
  PT_FLAG_FORCE_STREAM = 1u << 1, /* use the LDS-tile streaming kernel even when the scene fits in LDS */
  PT_FLAG_NO_FASTDIV = 1u << 2,   /* plain IEEE division for every rect/box side (no shared reciprocal) */
  PT_FLAG_TILE_GRANULAR = 1u << 3,  /* waves dequeue whole 8x8 tiles (default when the scene is LDS/scalar-cache resident) */
  PT_FLAG_FORCE_COOP = 1u << 7,     /* use the cooperative kernels even where the launcher's heuristic would not */
  PT_FLAG_NO_SPLIT = 1u << 8,       /* cooperative kernels, but no tile is rendered several lanes per pixel */
  PT_FLAG_NO_COOP = 1u << 6,        /* never split a ray's primitive list over idle lanes (tail acceleration off) */
  PT_FLAG_NO_LPT = 1u << 5,         /* skip the cost-probe pass: tiles are dequeued in raster order */
  PT_FLAG_PIXEL_GRANULAR = 1u << 4, /* lanes dequeue single pixels (default for the LDS-tile streaming kernel) */
  /* OPT-IN, NOT THE REFERENCE'S IMAGE: decorrelated RNG streams.  The reference gives a pixel ONE xorshift32 stream for all
   * of its samples (render.hpp:95-101,130-133), which makes a pixel a sequential chain and bounds any schedule by the
   * heaviest pixel.  With this flag a pixel's samples are cut into chunks of PT_FAST_CHUNK_SPP; chunk c of pixel id draws
   * from its own stream seeded pt_fast_seed(id, c), chunks are rendered as independent work units and their sums are added
   * in chunk order (deterministic run to run).  Same estimator, different random numbers: judged by PSNR / mean against
   * the parity image, never bit for bit, never the default.                                                             */
  PT_FLAG_FAST_RNG = 1u << 9,
  /* The reference's OTHER executor (render.hpp:113-122, built with USE_SINGLE_TASK for FPGA targets): ONE LocalPseudoRNG
   * with its default seed (xorshift.hpp:18) shared by all pixels, visited x-outer / y-inner, every draw of every sample of
   * every pixel taken from that one stream in order.  Sequential by definition — a pixel's first state depends on how
   * many numbers all earlier pixels drew — so it runs as ONE lane; offered for parity completeness on small frames
   * (width * height * samples <= 2^22), not for speed.  Needs shard_count == 1.                                         */
  PT_FLAG_SINGLE_STREAM = 1u << 10,
};

/* PT_FLAG_FAST_RNG: samples per chunk; pt_fast_seed() below gives the xorshift32 seed of chunk `chunk` of the pixel with
 * linear id `pixel`: h = pixel * 0x9E3779B1 + chunk * 0x85EBCA77 + 0x165667B1; h ^= h >> 16; h *= 0x7FEB352D; h ^= h >> 15;
 * h *= 0x846CA68B; h ^= h >> 16; 0 is mapped to 1 (a zero xorshift state is stuck).                                       */
#define PT_FAST_CHUNK_SPP 64

typedef struct PtRenderParams {
  int32_t width, height; /* template args of render<> (render.hpp:141)           */
  int32_t samples;       /* spp, template arg                                    */
  int32_t depth;         /* 50 in the reference (render.hpp:144)                 */
  int32_t shard_index;   /* this GPU's shard, 0 <= shard_index < shard_count     */
  int32_t shard_count;   /* 1 = whole frame                                      */
  uint32_t flags;
  int32_t reserved;
} PtRenderParams;

/* Opaque device-resident flattened scene.  Renders of one PtScene may be queued back to back (each launch gets its
 * own work-queue slot) but must be ORDERED with respect to each other — same stream, or synchronised by the caller —
 * and issued by one host thread at a time: the scene owns a small grow-only scheduling workspace (per-tile costs,
 * tile order) that consecutive renders reuse.  For concurrent renders create one PtScene per stream.            */
typedef struct PtScene PtScene;

/* ---- error codes (the reference returns void and asserts; we return codes) -- */
enum {
  PT_OK = 0,
  PT_ERR_INVALID_ARG = 1,
  PT_ERR_BAD_SCENE = 2,   /* tag / index out of range in the tables */
  PT_ERR_HIP = 3,         /* a HIP runtime call failed; see pt_last_error() */
  PT_ERR_NO_DEVICE = 4,
  PT_ERR_TOO_LARGE = 5
};

int pt_abi_version(void);
const char* pt_error_string(int code);
const char* pt_last_error(void); /* thread-local detail of the last failure */

/* camera(look_from, look_at, vup, vfov_deg, aspect, aperture, focus_dist, t0, t1)
 * camera.hpp:67-87.  Host-side arithmetic only.                                 */
int pt_camera_init(PtCamera* cam, const float look_from[3], const float look_at[3],
                   const float vup[3], float vfov_deg, float aspect_ratio,
                   float aperture, float focus_dist, float time0, float time1);

/* Validate + flatten + upload to the current HIP device.  Replaces the
 * sycl::buffer wrapping of hittables and image_texture::freeze()
 * (render.hpp:146-148).  Unlike freeze() it may be called any number of times. */
int pt_scene_create(const PtSceneDesc* desc, PtScene** out_scene);
void pt_scene_destroy(PtScene* scene);
/* Device memory the scene's DATA occupies: record blob + material table + the triangle pools' tables + texture atlas
 * (launch workspaces — tile orders, candidate caches — come on top and are sized by the frame).  -1 for NULL.      */
int64_t pt_scene_device_bytes(const PtScene* scene);
/* Which kernels this library was built from: the first 16 hex digits of the sha256 over its kernel sources
 * (csrc/Makefile) — what bench.py and tools/pmc_summary.py stamp PMC recordings with.                               */
const char* pt_build_id(void);

/* ---- tuning --------------------------------------------------------------------------------------------------------------
 * PERFORMANCE-ONLY knobs: every setting gives the same image bit for bit (tests/test_abi_cpu.py, the GPU parity suite runs
 * several of them); they choose which exact culling structures pt_scene_create builds and how launches are scheduled.  A
 * zero-initialised struct with struct_size set (pt_tuning_init) is the library's defaults.  pt_scene_create(desc, out) is
 * pt_scene_create_tuned(desc, NULL, out): the defaults with the PT_* environment variables applied on top — the override
 * channel of tools/ and of A/B runs (tools/README.md lists them).  A caller that passes a PtTuning gets EXACTLY that: the
 * environment is not consulted.  Fields (0 = default unless said otherwise):
 *   sphere_grid        -1: no culling grid for runs of small spheres (PT_NO_GRID)
 *   grid_margin/cell   the grid's margin in median radii / cell edge in (median radius + margin) (PT_GRID_M, PT_GRID_CELL; 0.5, 3.0)
 *   slab_pools         -1: no slab pools for rect / box stretches (PT_NO_BOXCULL); 1: a pool for every stretch of >= 2
 *                      (PT_POOL_ALWAYS: also where the cost model says it does not pay)
 *   tri_pool           -1: no triangle pool (PT_NO_TRICULL)
 *   tri_min_run        shortest triangle run that gets a pool (PT_TRI_MIN; 4096; PT_TRICULL=1 means 256)
 *   tri_M, tri_cell    the pool's slack 1/M (grid boxes grow with 1/M, bands with M) and its grid cell in median grown boxes (PT_TRI_M,
 *                      PT_TRI_CELL; 6, 0.30 — round 6, after the grid's slack was re-derived; 12, 0.22 before)
 *   tri_binned         scenes with ONE pooled triangle run render in GENERATIONS (round 6; csrc/pt_binned.hpp): every live pixel traces one
 *                      ray per generation and the rays are sorted by direction bin, so that a bin's list of grazing candidates is read once
 *                      for 64 rays — 1: on (PT_TRI_BINNED); 0: off: measured, it does not beat the persistent kernel yet (docs/EXPERIMENTS.md).
 *                      Such renders have returned only when the frame is done (the number of generations is known on the device only)
 *   sphere_merge       a sphere run of more than two spheres also tests, through its lists, the static spheres of later short sphere runs
 *                      that only rect / box / short triangle runs separate it from (the resident kernels then skip those runs: a short run
 *                      costs ~600 cycles per sphere, a list entry ~100) — 0: yes; -1: every run where it stands (PT_NO_SPHERE_MERGE; the A/B)
 *   tri_cache          triangle-pool kernels, pinhole cameras: a lane keeps the grazing candidates of its pixel's camera rays (built by the
 *                      pixel's first sample with the filters widened to the pixel's footprint) and every later camera ray of the pixel tests
 *                      those instead of enumerating its direction-map list — 0: yes; -1: no (PT_NO_TRI_CACHE; the A/B)
 *   tri_res[0..2], tri_rho[0..1] + tri_rho2   the pool's direction maps: resolution per cube-map face and the largest rho / R a map
 *                      serves, per rho class (PT_TRI_RES=a,b,c PT_TRI_RHO=a,b,c; {128, 64, 32} since round 6 — camera rays take their candidates from the pixel's cache: tri_cache —, {2.12, 4, 16}; a negative rho: no such
 *                      map; rays with rho beyond the last class stream every band record)
 *   tri_budget_mb      MiB the direction maps may take together (PT_TRI_BUDGET_MB; 1600, for all of a scene's pooled runs): a map over budget is built coarser or not at all
 *   generic_materials  1: no material-specialised kernels (PT_NO_MATSPEC)
 *   blocks_per_cu      cap on resident workgroups per CU (PT_BLOCKS_PER_CU)
 *   cold_state         -1: the cold lane state stays in registers (PT_NO_COLD_LDS)
 *   wide_log2_group    forced log2 group size of the cooperative kernels' wide phase (PT_WIDE_LOGG; 0: the model picks)
 *   split_tiles_mode / split_tiles   mode 1: `split_tiles` tiles go through the wide phase (< 0: all) (PT_SPLIT_TILES)
 *   lpt_by_max         1 / -1: order tiles by their heaviest pixel / by their ray count (PT_LPT_MAX; 0: by kernel family)
 *   probe_spp_max      depth cap of the cost probe (PT_PROBE_SPP_MAX; 16)
 *   grid_min_tiles     frames (shards) of fewer tiles keep the cooperative kernels and the lists (PT_GRID_MIN_TILES; 0)
 *   model_fixed/chain  constants of the makespan model (PT_MODEL_FIXED, PT_MODEL_CHAIN; 2400, 2400)
 *   scatter_log        triangle-pool kernels: log2 of the pixels of one tile a wave takes together (PT_SCATTER_LOG; 0)
 *   scatter_mode       -1: whole tiles per wave (PT_NO_SCATTER); 1: with the cost probe (PT_LPT_SCATTER)
 *   lanes_cap          sphere-grid kernels on frames that do not fill the chip: lanes of a wave that take pixels (PT_LANES_CAP=n; 0: 16 x
 *                      pixels per resident lane, whole tiles from 24 on; -1 or PT_LANES_CAP=0: whole tiles always)
 *   grid_walk          which sphere-grid walk: 1 wave-synchronous (each lane tests its candidate in place), 2 through the LDS pair queue
 *                      (64 pairs per batch); 0: the launcher's rule (PT_GRID_WALK)
 *   heavy_tiles        sphere-grid kernels on launches bound by their heaviest tiles' chains (1.5 ... 6 pixels per resident lane): the first
 *                      n tiles of the cost-sorted order are handed out 16 pixels at a time, a quarter tile per wave (PT_HEAVY_TILES=n;
 *                      0: one tile per SIMD of the chip in that range, none outside; -1: never)
 *   probe_resume       the cost probe's samples are the frame's first samples — the frame launch starts every pixel from the radiance sum and
 *                      the generator state the probe left (same stream, same order of additions) — 0: yes; -1: the probe's samples are
 *                      thrown away and rendered again, as before round 5 (PT_NO_PROBE_RESUME; the A/B)
 *   chain_priority     headline-family frame launches (no sphere grid, no cooperative phase): from the middle of the tile queue on, a wave
 *                      raises its issue priority by the samples its slowest pixel still has to render — the longest remaining chain first;
 *                      changes no value, only when a pixel is rendered — 0: yes; -1: never (PT_NO_CHAIN_PRIO; the A/B)                        */
typedef struct PtTuning {
  int32_t struct_size; /* sizeof(PtTuning) of the caller's header */
  int32_t sphere_grid;
  float grid_margin, grid_cell;
  int32_t slab_pools;
  int32_t tri_pool, tri_min_run;
  float tri_M;
  int32_t tri_binned; /* (round 6: took the place of tri_Mg, dead since round 5) */
  float tri_cell;
  int32_t tri_res[3];
  int32_t generic_materials;
  int32_t blocks_per_cu, cold_state, wide_log2_group, split_tiles_mode, split_tiles, lpt_by_max, probe_spp_max, grid_min_tiles;
  float model_fixed, model_chain;
  int32_t scatter_log, scatter_mode;
  int32_t lanes_cap, grid_walk;
  int32_t heavy_tiles;
  float tri_rho[2];      /* (round 5; these four took the place of reserved words: the struct's size is unchanged) */
  int32_t tri_budget_mb;
  float tri_rho2;
  int32_t probe_resume;  /* (round 5, the last reserved word) */
  int32_t chain_priority; /* (round 5: appended — a caller built against the shorter struct passes its own struct_size and gets the default) */
  int32_t tri_cache;      /* (round 6: appended) */
  int32_t sphere_merge;   /* (round 6: appended) */
} PtTuning;
void pt_tuning_init(PtTuning* t);     /* zero + struct_size: the library's defaults                                          */
void pt_tuning_from_env(PtTuning* t); /* the defaults with the PT_* environment applied: what pt_scene_create(desc, out) uses */
int pt_scene_create_tuned(const PtSceneDesc* desc, const PtTuning* tuning, PtScene** out_scene);

/* Number of floats the caller must provide to pt_render for these params:
 * shard_count==1: height*width*3 laid out [y][x][rgb], y=0 = bottom scan-line
 * (render.hpp:105, main.cpp:41).  shard_count>1: pt_shard_tiles()*64*3 laid
 * out [local_tile][ly*8+lx][rgb].                                              */
int64_t pt_framebuffer_floats(const PtRenderParams* p);
/* Seed of the fast mode's stream for (pixel, chunk) — see PT_FLAG_FAST_RNG; never 0. Host function, no GPU needed. */
uint32_t pt_fast_seed(uint32_t pixel, uint32_t chunk);

int32_t pt_shard_tiles(const PtRenderParams* p); /* ceil(n_tiles / shard_count) */

/* Optional: size the scene's per-launch workspaces (tile cost / order arrays and per-pixel generator states of the heaviest-first schedule; the fast
 * mode's partial sums) for these parameters NOW, so that no later pt_render() with parameters that need no more
 * allocates or frees device memory (hipMalloc / hipFree synchronise the device; without this call the first render at
 * a new size does it).  Call it with the scene's device current.                                                     */
int pt_scene_reserve(const PtScene* scene, const PtRenderParams* params);

/* The hot path.  Asynchronous on `stream` (a hipStream_t, or NULL for the
 * default stream) like queue.submit (render.hpp:151); fb_device is device
 * memory, fully overwritten (discard_write, render.hpp:152) — and READ BACK
 * while the render runs: the cost-probe pass leaves every pixel's radiance sum
 * of its first samples there and the frame launch carries on from it
 * (PtTuning.probe_resume), so the buffer must be ordinary readable device memory
 * and holds intermediate sums until the stream has passed the call.  The first
 * render of a scene at a new frame size grows the scene's workspaces (hipMalloc,
 * which synchronises); call pt_scene_reserve first where that matters.          */
int pt_render(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p,
              float* fb_device, void* stream);

/* Same, timed: brackets the kernel launch with HIP events recorded on `stream`
 * and returns the kernel's duration (blocks until done).                        */
int pt_render_timed(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p,
                    float* fb_device, void* stream, float* kernel_ms);

/* Convenience: render into host memory (allocates, renders, copies back, syncs). */
int pt_render_host(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p,
                   float* fb_host);

/* Root-side un-interleave after the RCCL gather: gathered is
 * [shard_count][pt_shard_tiles][64][3] device floats -> fb [height][width][3].  */
int pt_unshard_tiles(const float* gathered_device, const PtRenderParams* p,
                     float* fb_device, void* stream);

/* Output stage of main.cpp:33-59: sqrt gamma, clamp [0,0.999], *256 -> u8,
 * vertical flip; rgb8_device is [height][width][3], row 0 = top.               */
int pt_tonemap_rgb8(const float* fb_device, int32_t width, int32_t height,
                    uint8_t* rgb8_device, void* stream);

/* ---- function-level probes (parity tests call these; not used by render) ---- */

/* One iteration of the bounce loop render.hpp:58-89 per record: hit_world,
 * emitted, scatter (or sky).  n records in, n out; host pointers.              */
typedef struct PtBounceIn {
  float origin[3];
  float dir[3];
  float time;
  uint32_t rng_state;
  float attenuation[3];
} PtBounceIn;

enum { PT_BOUNCE_MISS = 0, PT_BOUNCE_SCATTERED = 1, PT_BOUNCE_ABSORBED = 2 };

typedef struct PtBounceOut {
  int32_t status;     /* PT_BOUNCE_* */
  int32_t hittable;   /* index of the nearest hittable, -1 on miss */
  int32_t material;   /* material index of the hit, -1 on miss */
  int32_t front_face;
  float t;
  float p[3];
  float normal[3];
  float u, v;         /* 0 unless the scene has an image texture */
  float color[3];     /* MISS: attenuation*sky; ABSORBED: emitted; SCATTERED: new attenuation */
  float sc_origin[3]; /* scattered ray (SCATTERED only) */
  float sc_dir[3];
  float sc_time;
  uint32_t rng_state; /* generator state after the bounce */
} PtBounceOut;

int pt_debug_bounce(const PtScene* scene, const PtBounceIn* in, PtBounceOut* out, int32_t n);

/* First camera ray of a pixel: render.hpp:96-99 + camera.hpp:93-100.
 * rng_state in/out; ray out.                                                   */
typedef struct PtCameraRay {
  float origin[3];
  float dir[3];
  float time;
  uint32_t rng_state;
} PtCameraRay;
int pt_debug_camera_rays(const PtCamera* cam, int32_t width, int32_t height,
                         const int32_t* xy /*[n][2]*/, const uint32_t* rng_in,
                         PtCameraRay* out, int32_t n);

/* Host-only view of what pt_scene_create() uploads (no GPU needed): the flattened blob
 * ([n_runs run headers (device kind, first record offset, count, first hittable)] then the
 * per-kind 16-byte records) and the material table (4 x 16 bytes each, texture inlined).
 * Pass NULL buffers to query the sizes.  flags_out: bit0 = has image texture, bit1 = has medium, bit2 = a triangle run
 * carries a triangle pool (exact culling tables for long runs of Moller-Trumbore triangles; csrc/pt_tripool.hpp).        */
int pt_debug_flatten(const PtSceneDesc* desc, float* blob_out, int64_t blob_cap_f4, int32_t* n_blob_f4,
                     int32_t* n_runs, float* mats_out, int64_t mats_cap_f4, int32_t* flags_out);
/* the same for an explicit PtTuning (NULL: defaults + environment, i.e. pt_debug_flatten) */
int pt_debug_flatten_tuned(const PtSceneDesc* desc, const PtTuning* tuning, float* blob_out, int64_t blob_cap_f4, int32_t* n_blob_f4,
                           int32_t* n_runs, float* mats_out, int64_t mats_cap_f4, int32_t* flags_out);

/* Host-only statistics of the triangle pool pt_scene_create() would build (no GPU needed): out[0] = triangles in pooled runs,
 * out[1] = triangles whose band covers every direction in the first map ("slivers"), out[2] = entries of the first direction map,
 * out[3] = of the other two together, in units of 1024 (0: not built), out[4] = the three maps' resolutions, 10 bits each, out[5] = 1000 x mean
 * grid cells per triangle, out[6] = blob size in 16-byte records (always), out[7] = spheres that sit in a sphere culling grid (always). */
int pt_debug_tri_pool(const PtSceneDesc* desc, int32_t out[8]);
/* The tables of the scene's triangle pools (host-only, like pt_debug_flatten): the second buffer pt_scene_create uploads beside the blob
 * (pt_flatten.hpp: put_tri_pool says what the pools' headers in the blob point at).  pool_out may be NULL to query the size (in 16-byte
 * records); tuning NULL = defaults + environment.                                                                                       */
int pt_debug_flatten_pool(const PtSceneDesc* desc, const PtTuning* tuning, float* pool_out, int64_t pool_cap_f4, int64_t* n_pool_f4);

/* What the scheduler decided for the LAST render of this scene (blocks until that render is done): out[0] = tiles sent
 * through the wide phase, out[1] = lanes per pixel there (0 when the render had no cost-probe pass or used a kernel
 * without that phase).  For tuning the makespan model (csrc/pt_render.hip: lpt_order_kernel) and for tests.            */
int pt_debug_schedule(const PtScene* scene, int32_t out[2]);

/* What the launcher decided for the LAST frame launch of this scene (csrc/pt_render.hip: launch): out[0] = workgroups launched,
 * out[1] = lanes of a wave that take pixels (PtTuning.lanes_cap's rule; 64 = whole tiles), out[2] = queue positions at the head of the
 * cost-sorted order that are handed out 16 pixels at a time (PtTuning.heavy_tiles' rule; 0 = none), out[3] = 1 if the sphere-grid walk
 * through the LDS pair queue was picked.  For tests of those rules.                                                                  */
int pt_debug_last_launch(const PtScene* scene, int32_t out[4]);

/* Device math used by the kernel, elementwise over host arrays.
 * op: 0 sin 1 cos 2 log 3 pow5 4 atan2(a,b) 5 asin 6 fmod(a,1) 7 sqrt 8 div(a,b)
 *     9 the shared-reciprocal exact quotient a/b used for rect/box sides (pt_device.hpp: div_exact)
 *     10 the reciprocal 1/a of the ray context, correctly rounded for 2^-40 <= |a| <= 2^40 (rcp_rn_guarded)
 *     11 the square root of a = 0 or 2^-60 <= a <= 4 as the RNG's unit_vec / in_unit_disk take it (sqrt_rn_unit)
 *     12 a / sqrt(b) as the sky of a regular ray takes unit_vector(d).y (b = d.d in [3 * 2^-80, 3 * 2^80]; sky_unit_y)
 *     13 the checker texture's decision `sin(a) sin(b) sin(1) < 0` as 1.0 / 0.0 (texture.hpp:43-45; checker_sines_negative)
 *     14 / 15 sin / cos through the fused form the unit-ball sampler uses (pt_math.hpp: sincosf_)                            */
int pt_debug_math(int32_t op, const float* a, const float* b, float* out, int64_t n);

/* The texel (column i, row j) an image texture of width x height texels and frequency freq selects on a sphere whose unit normal at the
 * hit is n (n_xyz: 3 floats per entry) — texture.hpp:140-157 over sphere.hpp:13-24.  out_ij = what the render kernels take (the texel
 * straight from the normal where that is provably unambiguous, the reference's chain otherwise: pt_device.hpp sphere_texel_fast),
 * exact_ij = the reference's chain alone, took_fast = 1 where the short form decided.  2 ints per entry.  uv4 (optional, 4 floats per
 * entry): (u, v) of the reference's chain, then (u, v) of the binary32 approximations — their deviation is what the short form's
 * margin must cover.  For tests.                                                                                                       */
int pt_debug_sphere_texel(const float* n_xyz, int64_t n, float freq, int32_t width, int32_t height, int32_t* out_ij, int32_t* exact_ij,
                          uint8_t* took_fast, float* uv4);

#ifdef __cplusplus
}
#endif
#endif /* PT_RENDER_H */
