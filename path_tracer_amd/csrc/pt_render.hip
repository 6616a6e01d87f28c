// pt_render.hip — kernels + C ABI (include/pt_render.h) of the MI355X-native render() hot path.
//
// Replaces render<W,H,S>() / executor() / render_pixel()  (reference include/render.hpp:25-160).
//
// Mapping: one lane owns one pixel for ALL of its samples — the reference's single xorshift32
// stream per pixel, consumed sequentially across samples (render.hpp:95-101,130-133), leaves no other bit-exact
// choice.  Lanes are persistent: they pull pixels (8x8 tiles by default) from a per-launch queue, heaviest tiles
// first (cost-probe pass + lpt_order_kernel), and run a "regenerate in place" loop — a lane whose path ended starts
// its next sample in the same iteration, so the wave-uniform primitive loop always runs with all live lanes (what
// ballot/prefix compaction would buy, without moving state between lanes).  Three kernel families:
//   render_kernel<IMG, LDS, MLDS, COOP>   scene resident in LDS (or the scalar cache); COOP adds the cooperative
//                                         traversal (a ray's list split over idle lanes) and the split queue
//   render_kernel_stream<IMG>             scene larger than LDS: workgroup-synchronous LDS-tile streaming
//   bounce / camera_rays / math kernels   function-level probes for the parity tests
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <type_traits>
#include <string>
#include <vector>

#include "../../include/pt_render.h"
#include "pt_device.hpp"
#include "pt_flatten.hpp"

#ifdef PT_STAMPS
__device__ unsigned long long g_stamps[8];
#endif
#ifdef PT_STAMPS_WALK
__device__ unsigned long long g_walk[8];
#endif
#ifdef PT_STAMPS_RUNS
__device__ unsigned long long g_runs[16];
#endif
#ifdef PT_STAMPS_TRI
__device__ unsigned long long g_tri[16];
#endif
#ifdef PT_STAMPS_BLOCKS /* diagnostic build (make variant NAME=libpt_blocks.so EXTRA=-DPT_STAMPS_BLOCKS; tools/block_residency.py): per workgroup of the
                           frame launch where it ran (XCC_ID << 32 | HW_ID) and when it started / ended (s_memrealtime, 100 MHz) */
__device__ unsigned long long g_blocks[3 * 8192];
#endif

using namespace ptd;

static_assert(sizeof(PtHittable) == 64 && sizeof(PtMaterial) == 32 && sizeof(PtTexture) == 48 && sizeof(PtCamera) == 96);
static_assert(sizeof(Cam) == sizeof(PtCamera));

namespace {

#ifndef PT_MIN_WAVES
#define PT_MIN_WAVES 7 /* non-IMG resident kernel: 71 VGPRs, no scratch; +6.5 % on the headline scene (A/B, one process) */
#endif
#ifndef PT_MIN_WAVES_IMG
#define PT_MIN_WAVES_IMG 5 /* image-texture kernels: 96 VGPRs; +10 % on the 496-hittable scene (A/B, one process) */
#endif
#ifndef PT_MIN_WAVES_CL
#define PT_MIN_WAVES_CL 7 /* cold lane state in LDS: 68 VGPRs; measured 8 waves (64 VGPRs, no spill) -5 %, 7 waves +2 %, 6 waves -3 % */
#endif
#ifndef PT_MIN_WAVES_COOP
#define PT_MIN_WAVES_COOP 5 /* cooperative kernels: 93 VGPRs, no scratch (7 waves: 72 VGPRs + 76 B/lane of spills in the loop) */
#endif
#ifndef PT_MIN_WAVES_TRIPOOL
#define PT_MIN_WAVES_TRIPOOL 5 /* triangle-pool kernels: 96 VGPRs.  Round 6, with the camera rays' candidate cache and two pair batches / two expansion trips in flight in the grid walk: 1080p x 32 spp 1 116 ms at 7 waves (72 VGPRs + 228 B of scratch), 986 at 6 (80 VGPRs), 978 at 5; with the pipelined walk 1 116 at 6, 979 at 5 — profiles/r06_ab_tri_cache.txt, r06_ab_tri_pipe.txt; rounds 3-5 (no cache, no pipelining) measured 7 best */
#endif
#ifndef PT_MIN_WAVES_BINSTEP
#define PT_MIN_WAVES_BINSTEP 4 /* bin_step_kernel (pt_binned.hpp): 128 VGPRs, no scratch (6 waves: 80 VGPRs + 236 B of scratch with the pipelined trips) */
#endif
#ifndef PT_MIN_WAVES_COOP_IMG
#define PT_MIN_WAVES_COOP_IMG 5 /* 96 VGPRs + 60 B/lane of spills; spill-free needs 116 VGPRs = 4 waves: 496-hittable scene -9 % (A/B) */
#endif
// How a kernel obtains the u,v an image texture looks up (DESIGN.md §3 "u,v of the final hit")
enum { UV_NONE = 0, UV_WINNER = 1, UV_TRACKED = 2 };
constexpr int kBlock = 256;                 // 4 wavefronts = 4 tiles per workgroup
constexpr int kWavesPerBlock = kBlock / 64;
#ifndef PT_GRID_BLOCK
#define PT_GRID_BLOCK 256
#endif
constexpr int kGridBlock = PT_GRID_BLOCK;   // workgroup size of the LDS-resident kernels that walk a sphere grid (render_kernel: BLOCK)
constexpr size_t kMaxLdsBlob = 64 * 1024;   // blob staged in LDS when it fits
constexpr size_t kMaxLdsWithMaterials = 16 * 1024; // stage the material table too when records + materials are this small
constexpr size_t kMaxLdsColdScene = 10 * 1024;    // records + materials this small: the 8-wave kernel with LDS-resident cold lane state
constexpr float kCoopMinTraversal = 2500.0f;        // estimated VALU instructions of one list scan (~110 spheres)
constexpr int kGridMinTiles = 0;                    // since the grid's retuning it wins at every frame size: see launch_render
constexpr unsigned kQueueRing = 256;        // launches in flight on one scene may not exceed this

struct KArgs {
  Cam cam;
  const f4* blob;
  const f4* mats;
  const f4* pool;      // tables of the triangle pools (pt_flatten.hpp: put_tri_pool); NULL when the scene has none
  const uint8_t* atlas;
  float* fb;
  int n_runs, blob_f4, mats_f4;
  int width, height, samples, depth;
  float inv_w, inv_h; // RN(1 / (float)width), RN(1 / (float)height): camera_ray
  int pinhole;        // cam_is_pinhole(cam): camera_ray skips the lens arithmetic
  int shard_index, shard_count;
  int tiles_x, n_tiles;
  int n_local_pixels;  // 64 x the tiles this shard owns (incl. padding pixels of edge tiles)
  unsigned int* queue; // per-launch dequeue counter, zeroed on the stream before the kernel
  unsigned int* cost;  // non-NULL: cost-probe pass, per local tile ray counts
  // The probe's samples are the frame's FIRST samples: with resume_rng the probe leaves every pixel's plain radiance sum in fb and its
  // generator's state in resume_rng[local pixel]; the frame launch (resume_spp = what the probe rendered) starts every pixel from both, at
  // sample resume_spp — the same stream, the same order of additions (render.hpp:95-101), and no sample is rendered twice.
  unsigned int* resume_rng;
  int resume_spp;
  // Headline-family launches (launch_render): once prio_onset queue positions are taken, a wave sets its issue priority — every 64
  // iterations — by the samples its slowest pixel still has to render: >= prio_t3 -> 3, >= prio_t2 -> 2, >= prio_t1 -> 1, else 0 (render_kernel).
  // prio_t1 = 0: never.
  int prio_onset, prio_t1, prio_t2, prio_t3;
  int cost_max;        // probe: 1 = keep the tile's HEAVIEST pixel (x 64) instead of the sum over its pixels
  const int* order;    // non-NULL: queue position -> local tile, heaviest first
  const int* n_split;  // non-NULL (COOP kernels): [0] how many leading tiles of `order` go through the wide phase, [1] log2 G
  int tile_granular;   // PT_FLAG_TILE_GRANULAR
  int tile_group;      // tile-granular launches: lanes of a wave that take pixels together (64: the whole wave = a tile; 32 / 16: half / quarter tiles)
  int heavy_pixels;    // > 0 (tile-granular grid kernels, chain-bound launches): the first heavy_pixels queue positions — the heaviest tiles of the
  int heavy_lanes;     //   cost-sorted order — are taken heavy_lanes pixels at a time, by the first heavy_lanes lanes of a wave (lane_acquire)
  int scatter_p;       // > 0 (triangle-pool kernels): queue positions are dealt to tiles in runs of 2^scatter_log pixels with this stride (lane_acquire)
  int scatter_log;
  int lanes_cap;       // < 64 (triangle-pool kernels, fewer pixels than lanes): only the first lanes_cap lanes of a wave take pixels (lane_acquire)
  int n_hittables;
  int coop_prefix;     // >= 0: cooperative traversal allowed, list splittable up to this hittable; -1: disabled
  int fast_ok; // every rect/box coordinate finite and <= 2^60: rays may use the shared-reciprocal division
  // PT_FLAG_FAST_RNG (opt-in, not the reference's image): a work unit is one CHUNK of a tile's pixels; `samples` above is
  // then the chunk length, fast_chunks the chunks per pixel, samples_total the caller's spp, fb the partial-sum workspace
  // [chunk][framebuffer layout] (fast_reduce_kernel adds the chunks in order and divides).  0 = the reference's single stream.
  int fast_chunks, samples_total;
  long long fast_stride; // floats per chunk plane of the workspace
  // the camera rays' candidate cache of the triangle-pool kernels (pt_device.hpp: TriPrimCtx): one line per resident lane; NULL: none
  unsigned int* tri_cache;
  float foot[10];        // (llc - origin) xyz, hor / W xyz, ver / H xyz, the bound dd on |d - d_centre| over a pixel's camera rays
};

// Per-lane state of the persistent loop.  A lane owns ONE pixel at a time, for all of its samples (the
// reference's single RNG stream per pixel, render.hpp:130-133, forbids splitting a pixel), but lanes are not
// tied to a tile: a lane that has finished its pixel pulls the next one from a per-launch queue, so every lane
// of the chip stays busy until the frame's pixels run out.
// The part of a lane's state that is touched only when a sample or a pixel ends — the radiance sum, the sample count,
// which pixel it is — lives either in registers or, for small scenes, in per-thread LDS slots (~10 LDS instructions per
// finished sample): 8 registers less pressure inside the traversal.  It would even fit 8 waves per SIMD (64 VGPRs, no
// spill), but that is slower than 7 (-5 % vs +2 % on the headline scene; PT_MIN_WAVES_CL).
typedef __attribute__((address_space(3))) float* lds_fp;
constexpr int kColdSlots = 8; // dwords per thread
template <bool IN_LDS> struct Cold;
template <> struct Cold<false> {
  V3 acc;
  int s, pix, x, y;
  unsigned int iters;
  __device__ __forceinline__ void init(lds_fp) { acc = mk(0.0f, 0.0f, 0.0f); s = 0; pix = -1; x = 0; y = 0; iters = 0; }
  __device__ __forceinline__ void begin(int pix_, int x_, int y_, int s0 = 0) { acc = mk(0.0f, 0.0f, 0.0f); s = s0; iters = 0; pix = pix_; x = x_; y = y_; }
  __device__ __forceinline__ void resume(V3 acc_, int s0) { acc = acc_; s = s0; }
  __device__ __forceinline__ int add_sample(V3 o) { acc = acc + o; return ++s; }
  __device__ __forceinline__ void count_ray() { iters++; }
  __device__ __forceinline__ V3 get_acc() const { return acc; }
  __device__ __forceinline__ int get_s() const { return s; }
  __device__ __forceinline__ int get_pix() const { return pix; }
  __device__ __forceinline__ int get_x() const { return x; }
  __device__ __forceinline__ int get_y() const { return y; }
  __device__ __forceinline__ unsigned int get_iters() const { return iters; }
};
template <> struct Cold<true> {
  lds_fp p; // this thread's slots: field k at p[k * kBlock]
  __device__ __forceinline__ void init(lds_fp base) { p = base + threadIdx.x; }
  __device__ __forceinline__ void begin(int pix_, int x_, int y_, int s0 = 0) {
    p[0] = 0.0f; p[kBlock] = 0.0f; p[2 * kBlock] = 0.0f;
    p[3 * kBlock] = __int_as_float(s0); p[4 * kBlock] = __int_as_float(pix_); p[5 * kBlock] = __int_as_float(x_);
    p[6 * kBlock] = __int_as_float(y_); p[7 * kBlock] = __int_as_float(0);
  }
  __device__ __forceinline__ void resume(V3 acc_, int s0) { p[0] = acc_.x; p[kBlock] = acc_.y; p[2 * kBlock] = acc_.z; p[3 * kBlock] = __int_as_float(s0); }
  __device__ __forceinline__ int add_sample(V3 o) {
    p[0] = p[0] + o.x; p[kBlock] = p[kBlock] + o.y; p[2 * kBlock] = p[2 * kBlock] + o.z;
    const int s = __float_as_int(p[3 * kBlock]) + 1;
    p[3 * kBlock] = __int_as_float(s);
    return s;
  }
  __device__ __forceinline__ void count_ray() { p[7 * kBlock] = __int_as_float(__float_as_int(p[7 * kBlock]) + 1); }
  __device__ __forceinline__ V3 get_acc() const { return mk(p[0], p[kBlock], p[2 * kBlock]); }
  __device__ __forceinline__ int get_s() const { return __float_as_int(p[3 * kBlock]); }
  __device__ __forceinline__ int get_pix() const { return __float_as_int(p[4 * kBlock]); }
  __device__ __forceinline__ int get_x() const { return __float_as_int(p[5 * kBlock]); }
  __device__ __forceinline__ int get_y() const { return __float_as_int(p[6 * kBlock]); }
  __device__ __forceinline__ unsigned int get_iters() const { return (unsigned int)__float_as_int(p[7 * kBlock]); }
};

template <bool CL>
struct LaneT {
  uint32_t rng;
  V3 att;
  Ray ray;
  int b;
  Cold<CL> cold; // radiance sum, sample count, pixel identity, ray count (cost probe)
  bool live;     // owns a pixel with samples left
  bool retired;  // the queue is empty for this lane
  bool need_new; // next iteration starts a new sample
  bool split_done; // (wave-uniform) the split queue is exhausted
  int wide;        // (wave-uniform) log2 lanes per pixel while this wave serves the split queue; 0 = one lane per pixel
  unsigned int split_pixels; // (wave-uniform) 64 x the tiles handed out through the split queue
};

template <typename Lane>
__device__ __forceinline__ void lane_reset(Lane& L, lds_fp cold_base) {
  L.rng = 0;
  L.att = mk(1.0f, 1.0f, 1.0f);
  L.ray.o = mk(0.0f, 0.0f, 0.0f); L.ray.d = mk(0.0f, 0.0f, 1.0f); L.ray.tm = 0.0f;
  L.b = 0;
  L.cold.init(cold_base);
  L.live = false; L.retired = false; L.need_new = true; L.split_done = false; L.wide = 0; L.split_pixels = 0;
}

// pt_fast_seed (include/pt_render.h), device and host
__host__ __device__ __forceinline__ uint32_t fast_seed(uint32_t pixel, uint32_t chunk) {
  uint32_t h = pixel * 0x9E3779B1u + chunk * 0x85EBCA77u + 0x165667B1u;
  h ^= h >> 16; h *= 0x7FEB352Du; h ^= h >> 15; h *= 0x846CA68Bu; h ^= h >> 16;
  return h ? h : 1u;
}

// Kernel arguments that only the rare paths read (queue, order, frame geometry, framebuffer ...): read through `a` they are
// SGPR-resident for the whole loop and push other values out into VGPR lanes (v_readlane + hazard nops in the hot path).  The
// rare paths therefore read them from the kernarg segment on the spot, through a pointer the optimiser cannot see through
// (so that the loads stay where they are).  The kernels take KArgs as their only argument: the segment IS a KArgs.
#ifdef __HIP_DEVICE_COMPILE__
#define PT_COLD_ARGS(name, a)                                                            \
  unsigned long long name##_p = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr(); \
  asm volatile("" : "+s"(name##_p));                                                      \
  const __attribute__((address_space(4))) KArgs& name = *(const __attribute__((address_space(4))) KArgs*)name##_p
#else
#define PT_COLD_ARGS(name, a) const KArgs& name = a
#endif

struct __attribute__((packed, aligned(4))) Rgb12 { float r, g, b; }; // a pixel of the framebuffer: global_{load,store}_dwordx3 need dword alignment only

// Wave-aggregated dequeue: one atomicAdd per wave for all lanes that need a pixel (ballot + prefix count),
// pixels handed out in tile order so a fresh wave starts on one coherent 8x8 tile.
//
// Wide phase (COOP kernels, L.wide = log2 G > 0): the heaviest tiles (the first *a.n_split positions of the cost-sorted
// order) come from a queue of their own, one PIXEL per aligned group of G lanes.  All G lanes of a group hold the same
// pixel and compute everything redundantly — same seed, same RNG draws, same shading — except the traversal, where each
// tests 1/G of the list (hit_world_lds) — so a pixel's sequential chain gets shorter without any state ever moving
// between lanes.  A wave leaves the phase when that queue is empty and its last wide pixel is done.
template <bool FAST = false, typename Lane>
__device__ __forceinline__ void lane_acquire(Lane& L, const KArgs& a) {
  // (triangle-pool kernels: a wave's time is the sum over its lanes — when a launch has fewer pixels than lanes, every wave takes
  // its share, lanes_cap pixels, instead of the first waves taking 64 each and the rest none)
  if (a.lanes_cap < 64 && (int)(threadIdx.x & 63) >= a.lanes_cap) L.retired = true;
  const bool want = !L.live && !L.retired;
  const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
  if (mask == 0) return;
  PT_COLD_ARGS(k, a);
  const int lane = threadIdx.x & 63;
  const int leader = __builtin_ctzll(mask);
  const unsigned int rank = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask, 0u));
  const unsigned int split_pixels = L.split_pixels; // read once per wave (render_kernel): constant during the launch
  unsigned int i = 0;
  if (L.wide && L.split_done) {                           // nothing left to hand out in this phase:
    if (__builtin_amdgcn_ballot_w64(L.live) != 0) return; //   idle groups wait for the wave's last wide pixels (no memory op)
    L.wide = 0;                                           //   all done: carry on with the ordinary queue below
  }
  if (L.wide) {
    const unsigned int groups = (unsigned int)__builtin_popcountll(mask) >> L.wide; // idle groups (all-or-none per group)
    unsigned int b = 0;
    if (lane == leader) b = atomicAdd(k.queue + 1, groups);
    b = __builtin_amdgcn_readlane(b, leader);
    if (b + groups >= split_pixels) L.split_done = true;
    i = b + (rank >> L.wide);
    const bool got = want && i < split_pixels;
    // wave-uniform decisions first (no ballot under a divergent branch)
    const bool any_got = __builtin_amdgcn_ballot_w64(got) != 0, any_live = __builtin_amdgcn_ballot_w64(L.live) != 0;
    if (!any_got && !any_live) L.wide = 0; // queue was already empty and nothing in flight: ordinary queue below
    else if (!got) return;                 // (groups that got a pixel go on to set it up)
  }
  if (!L.wide) {
    // tile-granular mode: a wave takes its next 64 pixels only when all of its lanes are idle — or (round 6, tile_group = 32 / 16) an aligned
    // GROUP of its lanes takes the next half / quarter tile as soon as that group is idle: a wave's iteration costs the same whether 64 or 40
    // of its lanes are live, and a tile's lanes idle from their own last sample to the tile's (lane utilisation 0.78 on the headline frame)
    unsigned long long mask2 = mask;
    if (k.tile_granular) {
      const unsigned long long live_m = __builtin_amdgcn_ballot_w64(L.live);
      if (k.tile_group >= 64) { if (live_m != 0) return; }
      else {
        const int g0 = lane & ~(k.tile_group - 1);
        const unsigned long long gm = ((k.tile_group == 32 ? 0xffffffffull : 0xffffull) << g0);
        const bool group_idle = (live_m & gm) == 0;
        mask2 = __builtin_amdgcn_ballot_w64(want && group_idle);
        if (mask2 == 0) return;
      }
    }
    // The heaviest tiles in narrower waves.  A tile's time is its heaviest pixel's sequential chain times the wave's iteration, and an
    // iteration of a wave that steps 16 pixels together is shorter than one that steps 64 (the longest grid walk, the largest candidate
    // count, every material among them): the 4K frame's heaviest tiles alone take 55 / 48 / 43 ms at 64 / 32 / 16 lanes.  A launch that is
    // bound by those chains (launch_render: heavy_pixels) hands the head of the cost-sorted queue out heavy_lanes pixels at a time — a
    // quarter of a tile per wave, the other lanes idle until it is done — and everything behind it as whole tiles.  (The peek races with
    // other waves' takes: a wave may take a narrow piece just behind the head, or a whole tile just inside it; both are merely other
    // partitions of the same pixels.  At launch every resident wave peeks an untouched queue, so the head is never shorter than one piece per
    // wave — a quarter of the resident waves in tiles, which is the rule's own length; shorter forced heads measure the same.)
    const bool want2 = want && ((mask2 >> lane) & 1ull);
    const int leader2 = __builtin_ctzll(mask2);
    const unsigned int rank2 = __builtin_amdgcn_mbcnt_hi((unsigned int)(mask2 >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)mask2, 0u));
    unsigned int take = (unsigned int)__builtin_popcountll(mask2);
    if (k.heavy_pixels > 0 && k.tile_granular) {
      unsigned int cur = 0;
      if (lane == leader2) cur = __hip_atomic_load(k.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      cur = __builtin_amdgcn_readlane(cur, leader2);
      if (cur < (unsigned int)k.heavy_pixels) take = min(take, (unsigned int)k.heavy_lanes);
    }
    unsigned int base = 0;
    if (lane == leader2) base = atomicAdd(k.queue, take);
    base = __builtin_amdgcn_readlane(base, leader2);
    if (!want2 || rank2 >= take) return;
    i = split_pixels + base + rank2;
  }
  if (i >= (unsigned int)k.n_local_pixels) { L.retired = true; return; }
  // fast mode: the queue hands out (tile, chunk) units, a tile's chunks back to back
  unsigned int unit = i >> 6;
  int in_tile = (int)(i & 63);
  {
    // Triangle-pool kernels: a wave takes its live rays ONE AT A TIME, so its time is the SUM over its lanes' pixels, and 64 pixels of
    // one heavy tile in one wave are a chain 64 times as long as the heaviest pixel's (a shard of 1/8 of the 100 k-triangle frame took
    // 3.3 s of the whole frame's 3.8 s).  So consecutive queue positions are dealt in runs of 2^scatter_log pixels (default: single
    // pixels) to tiles that lie scatter_p apart in the (cost-sorted) order: a wave's 64 pixels come from 64 tiles spread evenly over the
    // cost distribution — every wave gets a stratified sample of the frame's work — and a tile's pixels go to 64 different waves.
    // 1080p x 32 spp: whole frame 3.78 -> 3.12 s, shard 0/8 3.29 -> 0.53 s (runs of 2 / 4 / 8 / 16 pixels: 0.59 / 0.65 / 0.82 / 1.22 s).
    // A bijection of [0, 64 nt): run c = i >> g -> (position c mod nt, run-in-tile (c / nt) mod (64 >> g)), position -> position *
    // scatter_p mod nt with gcd(scatter_p, nt) = 1.  Which pixel a lane renders changes nothing in the image: a pixel's seed is its id.
    if (k.scatter_p > 0) {
      const unsigned int g = (unsigned int)k.scatter_log, nt = (unsigned int)k.n_local_pixels >> 6, c = i >> g;
      const unsigned int pos = c % nt, row = (c / nt) & ((64u >> g) - 1u);
      unit = (unsigned int)(((unsigned long long)pos * (unsigned int)k.scatter_p) % nt);
      in_tile = (int)((row << g) + (i & ((1u << g) - 1u)));
    }
  }
  int chunk = 0;
  if constexpr (FAST) { chunk = (int)(unit % (unsigned int)k.fast_chunks); unit /= (unsigned int)k.fast_chunks; } // (fast mode: nt counts (tile, chunk) units)
  // queue position -> local tile: identity, or the cost-sorted order of the probe pass (heaviest tiles first)
  const int l = k.order ? k.order[unit] : (int)unit;
  const long long g = (long long)l * k.shard_count + k.shard_index; // global tile (pt_render.h: round-robin shards)
  const int tx = (int)(g % k.tiles_x), ty = (int)(g / k.tiles_x);
  const int x = tx * PT_TILE + (in_tile & 7), y = ty * PT_TILE + (in_tile >> 3);
  if (g >= k.n_tiles || x >= k.width || y >= k.height) return; // padding pixel: stays 0, ask again next iteration
  // render.hpp:130-132: seed = linear id of the pixel in the WHOLE frame, truncated to 32 bits
  const uint32_t id = (uint32_t)((unsigned long long)y * (unsigned long long)k.width + (unsigned long long)x);
  if constexpr (FAST) {
    // the last chunk of a pixel may be shorter: its sample counter starts ahead so that every chunk ends at k.samples
    const int n_here = min(k.samples, k.samples_total - chunk * k.samples);
    L.cold.begin((l * PT_TILE_PIXELS + in_tile) | (chunk << 24), x, y, k.samples - n_here);
    L.rng = fast_seed(id, (uint32_t)chunk);
  } else {
    L.cold.begin(l * PT_TILE_PIXELS + in_tile, x, y);
    L.rng = id;
    if (k.resume_spp > 0) { // the probe pass rendered this pixel's first samples (KArgs.resume_rng): carry on where it stopped
      const int pix = l * PT_TILE_PIXELS + in_tile;
      const long long at = k.shard_count == 1 ? ((long long)y * k.width + x) * 3 : (long long)pix * 3;
      const Rgb12 part = *(const Rgb12*)(k.fb + at);
      L.cold.resume(mk(part.r, part.g, part.b), k.resume_spp);
      L.rng = k.resume_rng[pix];
    }
  }
  L.live = true;
  L.need_new = true;
}

// One 12-byte store per finished pixel (render.hpp:105): global_store_dwordx3 needs dword alignment only.
// (HBM write traffic of a frame launch: 26.3 MB against the algorithmic 24.9 MB — profiles/r03_cornell_pmc_write.csv, the
// 148 ms dispatch.  The "68 MB" of the round-2 summaries was the cost-probe launch's per-pixel atomics, picked up because the
// summary took each counter's maximum over the two launches: tools/pmc_summary.py now reads the frame launch only.  Deferring
// the stores until a wave's whole tile is done — eight lanes writing 96 contiguous bytes — was tried and changes nothing:
// 26.26 MB.)
__device__ __forceinline__ void store_rgb(float* p, V3 c) { *(Rgb12*)p = Rgb12{c.x, c.y, c.z}; }

template <bool FAST = false, typename Lane>
__device__ __forceinline__ void lane_store(Lane& L, const KArgs& a) {
  L.live = false;
  PT_COLD_ARGS(k, a);
  if (L.wide && ((threadIdx.x & 63) & ((1 << L.wide) - 1))) return; // wide phase: one lane of the group writes
  if (k.cost) { // cost-probe pass: only the tile's ray count is kept
    // a wave holds a tile until its last pixel is done, so a tile's duration follows its heaviest pixel; the cooperative
    // kernels' model also needs the lane time, i.e. the sum
    if (k.cost_max) atomicMax(&k.cost[L.cold.get_pix() >> 6], L.cold.get_iters() * PT_TILE_PIXELS);
    else atomicAdd(&k.cost[L.cold.get_pix() >> 6], L.cold.get_iters());
    if (k.resume_rng) { // ... and, where the frame launch resumes from it, the pixel's state after these samples (KArgs.resume_rng)
      const long long at = k.shard_count == 1 ? ((long long)L.cold.get_y() * k.width + L.cold.get_x()) * 3 : (long long)L.cold.get_pix() * 3;
      store_rgb(k.fb + at, L.cold.get_acc());
      k.resume_rng[L.cold.get_pix()] = L.rng;
    }
    return;
  }
  long long idx;
  if constexpr (FAST) { // fast mode: this chunk's plain sum into its plane of the workspace
    const int packed = L.cold.get_pix(), pix = packed & 0xffffff;
    if (k.shard_count == 1) idx = ((long long)L.cold.get_y() * k.width + L.cold.get_x()) * 3;
    else idx = (long long)pix * 3;
    idx += (long long)(packed >> 24) * k.fast_stride;
    store_rgb(k.fb + idx, L.cold.get_acc());
    return;
  }
  V3 acc = L.cold.get_acc() / (float)k.samples; // render.hpp:102
  if (k.shard_count == 1) idx = ((long long)L.cold.get_y() * k.width + L.cold.get_x()) * 3;
  else idx = (long long)L.cold.get_pix() * 3;
  store_rgb(k.fb + idx, acc);
}

// Start the next sample of a lane whose path ended (render.hpp:95-99); the pixel itself is finished where its last
// sample ends (lane_shade), so this — and the camera code — appears once in the loop.
template <typename Lane>
__device__ __forceinline__ void lane_regenerate(Lane& L, const KArgs& a) {
  if (L.live && L.need_new) {
    // The camera is 30 floats of kernel argument.  Read through `a.cam` they sit in SGPRs for the whole loop, and the
    // register allocator, out of SGPRs, parks others in VGPR lanes (v_writelane / v_readlane + hazard s_nops all over the
    // loop).  KArgs starts with the camera, so here it is read from the kernarg segment where it is needed: four scalar
    // loads per regeneration instead.  (The pointer is made opaque so that the loads are not hoisted out of the loop.)
    static_assert(offsetof(KArgs, cam) == 0, "the camera leads the kernel arguments");
#ifdef __HIP_DEVICE_COMPILE__
    typedef const __attribute__((address_space(4))) Cam* kcam_p;
    unsigned long long kp = (unsigned long long)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(kp));
    const Cam cam = *(kcam_p)kp;
    const __attribute__((address_space(4))) KArgs& k = *(const __attribute__((address_space(4))) KArgs*)kp; // frame geometry: same reasoning
#else
    const Cam cam = a.cam; // (host pass of the single-source compile: never run)
    const KArgs& k = a;
#endif
    L.ray = camera_ray(cam, L.cold.get_x(), L.cold.get_y(), k.width, k.height, k.inv_w, k.inv_h, L.rng, k.pinhole != 0);
    L.att = mk(1.0f, 1.0f, 1.0f);
    L.b = 0;
    L.need_new = false;
  }
}

// The texture coordinates of a lane's hit, asked for by image textures only (pt_device.hpp: texture_value): (u, v) as the reference
// defines them, and — for kernels that derive them from the final hit — the sphere's normal itself, from which the texel follows
// without the mercator pair wherever that is unambiguous (pt_device.hpp: sphere_texel_fast).
template <int UV, typename PB>
struct HitUv {
  PB recs;
  const HitState& h;
  const Ray& ray;
  const Rec& rec;
  __device__ __forceinline__ void operator()(float& u, float& v) const {
    if (UV == UV_TRACKED) { u = h.u; v = h.v; }
    else if (UV == UV_WINNER) winner_uv(recs, h.hit, ray, h.closest, rec, u, v);
    else { u = 0.0f; v = 0.0f; }
  }
  __device__ __forceinline__ bool sphere_normal(V3& n) const {
    if constexpr (UV == UV_WINNER) {
      if (hit_kind(h.hit) == DK_SPHERE) { n = rec.normal; return true; } // winner_uv: mercator(rec.normal)
    }
    return false;
  }
};

// emitted / scatter / sky for the nearest hit (render.hpp:60-88) and the sample bookkeeping (:100).
template <int UV, bool FAST = false, int MATS = MATS_ALL, typename Lane, typename PB, typename PM>
__device__ __forceinline__ void lane_shade(Lane& L, const KArgs& a, const HitState& h, PB recs, PM mats, bool regular = false) {
  if (!L.live) return;
  if (a.cost) L.cold.count_ray(); // cost-probe pass (wave-uniform)
  V3 out = mk(0.0f, 0.0f, 0.0f);
  bool cont;
  if (h.hit < 0) {
    out = sky_color(L.ray, L.att, regular);
    cont = false;
  } else {
    Rec rec = resolve_hit<(MATS & MATS_RECTBOX_ONLY) != 0>(recs, h.hit, L.ray, h.closest); // per-lane gather of the one record that was hit
    // UV_TRACKED kernels carried u,v through the scan (stale values included); UV_WINNER derives them from the final hit
    // when an image texture asks; UV_NONE: the scene has no image texture, nothing reads them
    const HitUv<UV, PB> uv{recs, h, L.ray, rec};
    cont = shade<(MATS & 0x1ff)>(mats, a.atlas, rec, uv, L.ray, L.att, L.rng, out, regular);
    if (cont && ++L.b >= a.depth) { // bounce loop exhausted: black (render.hpp:91)
      out = mk(0.0f, 0.0f, 0.0f);
      cont = false;
    }
  }
  if (!cont) {
    L.need_new = true;
    // final_color += get_color(r)  render.hpp:100; pixel done: the lane is idle from here on
    if (L.cold.add_sample(out) == a.samples) lane_store<FAST>(L, a);
  }
}

// One turn of the crank before tracing: finish/advance pixels, pull new ones, start new samples.
// Returns false when this lane has nothing to trace this iteration.
template <bool FAST = false, typename Lane>
__device__ __forceinline__ void lane_prepare(Lane& L, const KArgs& a) {
  lane_acquire<FAST>(L, a); // idle lanes pull their next pixel ...
  lane_regenerate(L, a); // ... and every lane whose path ended (or that is new) starts a sample
}

#ifdef PT_STAMPS
// Diagnostic build only (`make stamps` -> libpt_stamps.so; never shipped, never timed): s_memtime shares of one loop
// iteration of the resident kernels, summed over waves; read with pt_debug_stamps / tools/stamps.py.
#define PT_STAMP(var) unsigned long long var = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_sched_barrier(0)
#endif

// Scene blob resident for the whole kernel: staged once into LDS (LDS=true) or read through the scalar cache.
// MLDS: the material table is staged too (small tables only: it rides behind the records in the same buffer), so
// the whole bounce — traversal, hit record, material, texture constants — runs out of LDS without a global load.
// COOP: the traversal can split a ray's list over idle lanes (hit_world_lds); costs ~10 VGPRs and ~7 % of the
// ordinary-mode throughput, so the launcher picks it only where the makespan floor matters (launch_render).
// CL: the cold part of the lane state lives in LDS (Cold<true>); small scenes only (8 KB per workgroup).
// FAST: PT_FLAG_FAST_RNG (opt-in decorrelated mode; its own instantiations, so the parity kernels carry none of it)
// BADOUEL: the scene has Badouel-strategy triangles (their loop is compiled only into these instantiations)
// GRID: the kernel carries the sphere-grid walk (pt_device.hpp: sphere_grid_walk).  Scenes without a culling grid and without
// an image texture — the headline Cornell-style scene — run instantiations without it (less code in the hot kernel, and
// nothing the walk needs can weigh on its 72-register budget).
// TRIPOOL: the kernel carries the exact culling of long triangle runs (pt_device.hpp: tri_pool_scan); scalar-cache variant only
// (such scenes are far beyond LDS), its own register budget.
// MATS: the material / texture kinds the scene can contain (pt_device.hpp: MATS_*): the headline family has instantiations for
// "lambertian + lightsource over solid textures" that carry no metal / glass / isotropic / checker / image code.
// BLOCK: threads per workgroup.  The loop has no barrier, so the workgroup size only decides how many waves share one LDS image of the scene:
// the 496-hittable scene's 31 KB image + the queued walk's per-wave LDS (1 KB) fit four workgroups per CU, i.e. four waves per SIMD at 256
// threads and five at 320 (kGridBlock).
template <int UV, bool LDS, bool MLDS, bool COOP, bool CL = false, bool FAST = false, bool BADOUEL = false, int GRID = 1, bool TRIPOOL = false,
          int MATS = MATS_ALL, int BLOCK = kBlock>
__global__ __launch_bounds__(BLOCK, TRIPOOL ? PT_MIN_WAVES_TRIPOOL : CL ? PT_MIN_WAVES_CL : COOP ? (UV ? PT_MIN_WAVES_COOP_IMG : PT_MIN_WAVES_COOP) : (UV ? PT_MIN_WAVES_IMG : PT_MIN_WAVES))
void render_kernel(KArgs a) {
  constexpr bool IMG = UV == UV_TRACKED;
  typedef LaneT<CL> Lane;
  static_assert(!CL || BLOCK == kBlock, "the LDS-resident cold lane state is laid out for kBlock threads");
  static_assert(BLOCK % 64 == 0 && BLOCK / 64 <= PT_MAX_WAVES_PER_BLOCK, "per-wave LDS arrays are sized for PT_MAX_WAVES_PER_BLOCK waves");
  static_assert(!TRIPOOL || BLOCK == kBlock, "the triangle pool's per-wave LDS arrays are sized for kBlock threads");
  __shared__ float cold_slots[CL ? kColdSlots * kBlock : 1];
  extern __shared__ f4 smem[];
  if (LDS) {
    const int n = a.blob_f4 + (MLDS ? a.mats_f4 : 0); // a.mats == a.blob + a.blob_f4 (one device buffer)
    for (int i = threadIdx.x; i < n; i += BLOCK) smem[i] = a.blob[i];
    __syncthreads();
  }
#ifdef PT_STAMPS_WALK
  if (threadIdx.x < 8) walk_ctr()[threadIdx.x] = 0;
  __syncthreads();
#endif
#ifdef PT_STAMPS_BLOCKS
  if (threadIdx.x == 0 && !a.cost && blockIdx.x < 8192) {
    const unsigned long long where = ((unsigned long long)__builtin_amdgcn_s_getreg(0x1814) << 32) | (unsigned int)__builtin_amdgcn_s_getreg(0xF804); // XCC_ID, HW_ID
    g_blocks[3 * blockIdx.x] = where;
    g_blocks[3 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime();
  }
  struct BlockEnd { bool on; unsigned int b; __device__ ~BlockEnd() { if (on) g_blocks[3 * b + 2] = __builtin_amdgcn_s_memrealtime(); } } block_end{threadIdx.x == 0 && !a.cost && blockIdx.x < 8192, blockIdx.x};
#endif
  Lane L;
  lane_reset(L, (lds_fp)cold_slots);
  if (COOP && a.n_split) {
    L.split_pixels = (unsigned int)a.n_split[0] * PT_TILE_PIXELS;
    L.wide = a.n_split[1];
  }
  if (a.depth <= 0) return; // depth 0: every sample returns black (render.hpp:58,91); the frame is pre-zeroed
#ifdef PT_STAMPS
  unsigned long long s_prep = 0, s_trav = 0, s_shade = 0, s_iters = 0;
#endif
  unsigned int prio_it = 0; // (headline family) iterations of this wave, for the priority poll below
  for (;;) {
#ifdef PT_STAMPS
    PT_STAMP(t0);
#endif
    // LONGEST REMAINING CHAIN FIRST, from mid-frame on (headline family only: kernels without a grid walk, a cooperative phase or a pool).
    // All resident waves take the heaviest tiles at t = 0, and a third of them never take another: their tile runs for the whole frame,
    // at 3.5 x a lone wave's iteration because six other waves share the SIMD's issue slots — and the frame ends with the longest of those
    // chains, ~10 ms after its work would be done if it could be spread (profiles/r05_wave_tail_cornell.txt).  A pixel's chain cannot be
    // cut, but its wave can be given the slots: once half the queue is taken, every wave tells the SIMD's arbiter how much its slowest
    // pixel still has to do — more than 1/2, 1/4, 1/8 of the samples: priority 3, 2, 1 (s_setprio; polled every 64 iterations: one scalar
    // compare per iteration, ~25 instructions per poll).  Waves on light tiles lose slots they had to spare; the long chains end earlier:
    // Cornell-style 1080p x 1024 spp 139.5 -> 128.2 ms, x 256 spp 37.8 -> 34.9, shard 0 of 4 / 8: 51.9 / 41.6 -> 45.6 / 40.3
    // (profiles/r05_ab_chain_priority.txt; ladders 2.5-5-10, 3-6-12, 2-4-16 and onsets 7/16 ... 9/16 within 1 %; from the start of the
    // frame, or by the PROBE's estimate of a tile's heaviest pixel: worse than none — what a wave knows about its own progress is exact, what the
    // probe guessed is not).  The grid kernels lose 5 % with it (latency-bound at four waves per SIMD) and do not carry it.
    if constexpr (GRID == 0 && !COOP && !FAST && !TRIPOOL && LDS) {
      if (a.prio_t1 > 0 && ((++prio_it) & 63u) == 0u) {
        PT_COLD_ARGS(k, a);
        unsigned int head = 0;
        if ((threadIdx.x & 63) == 0) head = __hip_atomic_load(k.queue, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        head = __builtin_amdgcn_readfirstlane(head);
        if (head >= (unsigned int)k.prio_onset) {
          int rem = L.live ? k.samples - L.cold.get_s() : 0;
#pragma unroll
          for (int st = 32; st >= 1; st >>= 1) rem = max(rem, __shfl_xor(rem, st, 64));
          if (rem >= k.prio_t3) __builtin_amdgcn_s_setprio(3);
          else if (rem >= k.prio_t2) __builtin_amdgcn_s_setprio(2);
          else if (rem >= k.prio_t1) __builtin_amdgcn_s_setprio(1);
          else __builtin_amdgcn_s_setprio(0);
        }
      }
    }
    lane_prepare<FAST>(L, a);
    if (__builtin_amdgcn_ballot_w64(L.live) == 0) {
      if (__builtin_amdgcn_ballot_w64(!L.retired) == 0) break; // queue drained for the whole wave
      continue;                                                 // only padding pixels this time: pull again
    }
#ifdef PT_STAMPS
    PT_STAMP(t1);
#endif
    HitState h;
    bool regular = false; // (wave-uniform) every live ray is regular: the sky's shortcut (sky_unit_y)
    if constexpr (LDS) {
      if constexpr (COOP) {
        // ordinary and cooperative traversal (few live lanes: each live ray's list split over the idle lanes)
        const CoopScene cs{a.n_runs, a.coop_prefix};
        hit_world_lds<IMG>((lds_f4p)smem, (cst_f4p)a.blob, cs, L.ray, L.rng, L.live, true, a.fast_ok != 0, L.wide, h);
      } else {
        RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
        c.live = L.live;
        const bool fast = wave_all_regular(c, L.live);
        regular = fast;
        hit_world<IMG, BADOUEL, GRID, false, (MATS & MATS_RECTBOX_ONLY) != 0>((lds_f4p)smem, (cst_f4p)a.blob, a.n_runs, c, fast, L.rng, h);
      }
#ifdef PT_STAMPS
      asm volatile("" ::"v"(h.closest), "v"(h.hit));
      PT_STAMP(t2);
#endif
      if constexpr (MLDS) lane_shade<UV, FAST, MATS>(L, a, h, (lds_f4p)smem, (lds_f4p)smem + a.blob_f4, regular);
      else lane_shade<UV, FAST, MATS>(L, a, h, (lds_f4p)smem, a.mats, regular);
#ifdef PT_STAMPS
      asm volatile("" ::"v"(L.att.x), "v"(L.ray.d.x));
      PT_STAMP(t3);
      s_prep += t1 - t0; s_trav += t2 - t1; s_shade += t3 - t2; s_iters++;
#endif
    } else {
      RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
      c.live = L.live;
      const bool fast = wave_all_regular(c, L.live);
      if constexpr (TRIPOOL) {
        TriPrimCtx pc;
        pc.cache = a.tri_cache;
        pc.prim = L.live && L.b == 0;
        pc.xy = L.cold.get_x() | (L.cold.get_y() << 16);
        pc.pix = L.cold.get_pix();
#ifdef __HIP_DEVICE_COMPILE__
        pc.foot = (const __attribute__((address_space(4))) float*)((unsigned long long)__builtin_amdgcn_kernarg_segment_ptr() + offsetof(KArgs, foot));
#else
        pc.foot = nullptr;
#endif
        hit_world<IMG, BADOUEL, GRID, TRIPOOL, (MATS & MATS_RECTBOX_ONLY) != 0>((cst_f4p)a.blob, (cst_f4p)a.blob, a.n_runs, c, fast, L.rng, h, a.pool, &pc);
      } else
      hit_world<IMG, BADOUEL, GRID, TRIPOOL, (MATS & MATS_RECTBOX_ONLY) != 0>((cst_f4p)a.blob, (cst_f4p)a.blob, a.n_runs, c, fast, L.rng, h, a.pool);
      lane_shade<UV, FAST, MATS>(L, a, h, a.blob, a.mats, fast);
    }
  }
#ifdef PT_STAMPS_WALK
  __syncthreads();
  if (threadIdx.x < 8) atomicAdd(&g_walk[threadIdx.x], walk_ctr()[threadIdx.x]);
#endif
#ifdef PT_STAMPS
  if ((threadIdx.x & 63) == 0) {
    atomicAdd(&g_stamps[0], s_prep); atomicAdd(&g_stamps[1], s_trav); atomicAdd(&g_stamps[2], s_shade);
    atomicAdd(&g_stamps[3], s_iters);
  }
#endif
}

// Scene blob larger than LDS (100 k triangles = 4.8 MB): the workgroup walks the list in lock-step and streams
// each run through one LDS tile — cooperative 16 B/lane copy, barrier, every wave tests its 64 rays against the
// tile's records by LDS broadcast, barrier.  Arithmetic intensity is ~200 lane-ops per streamed byte
// (256 rays x ~40 ops per 48-byte triangle), so the stream needs < 0.4 TB/s chip-wide: compute-bound by design.
constexpr int kTileF4 = 2016;      // 32,256 B; a multiple of every record size (2, 3, 4 f4); five workgroups per CU
constexpr int kSmallRunF4 = 48;    // runs this short are read through the scalar cache instead (no barriers)

// Tail of such a frame: a pixel is one sequential chain and a full scan of the list is long (100 k triangles: ~15 ms per
// workgroup iteration), so the heaviest pixels (8-15 rays per sample against a mean of 2) keep a few lanes busy long after
// the queue is empty, while the lock-step scan costs the same whether 1 or 64 lanes of a wave are live.  So a wave that is
// down to <= 32 live rays spreads each of them over G = 64 >> ceil(log2 live) lanes for the whole scan (lane j of a group
// tests every G-th record of each tile; one butterfly merge with the reference's acceptance rule at the end: see
// hit_world_lds), and a wave with no live ray only keeps the barriers.  Scenes with a constant_medium (in-traversal RNG
// draw) or stale-u,v hazards scan the ordinary way.
template <int UV, bool FAST = false, bool BADOUEL = false>
__global__ __launch_bounds__(kBlock) void render_kernel_stream(KArgs a) {
  constexpr bool IMG = UV == UV_TRACKED;
  __shared__ f4 tile[kTileF4];
  typedef LaneT<false> Lane;
  Lane L;
  lane_reset(L, (lds_fp) nullptr);
  if (a.depth <= 0) return;
  const cst_f4p cblob = (cst_f4p)a.blob;
  const bool coop_scene = !IMG && a.coop_prefix >= a.n_hittables; // list splittable end to end (no medium)
  for (;;) {
    lane_prepare<FAST>(L, a);
    if (!__syncthreads_or(L.live)) {
      if (!__syncthreads_or(!L.retired)) break;
      continue;
    }
    RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
    c.live = L.live;
    const bool fast = wave_all_regular(c, L.live);
    const unsigned long long live_mask = __builtin_amdgcn_ballot_w64(L.live);
    const int nlive = __builtin_popcountll(live_mask);
    const bool wave_idle = nlive == 0; // nothing to trace: this wave only copies tiles and keeps the barriers
    int logG = 0;
    if (coop_scene && fast && nlive >= 1 && nlive <= 32) {
      logG = coop_group_log(nlive);
      coop_handoff(c, live_mask, nlive, logG);
      c.live = ((threadIdx.x & 63) >> logG) < nlive; // lanes of a group that serves a live ray
    }
    const int j = (threadIdx.x & 63) & ((1 << logG) - 1);
    HitState h;
    hit_begin(h);
    for (int ri = 0; ri < a.n_runs; ++ri) {
      f4 runf = cblob[ri];
      const int kind = as_i(runf.x) & 7, off = as_i(runf.y), cnt = as_i(runf.z); // (& 7: this kernel scans every run where it stands, absorbed or not — pt_flatten.hpp)
      const int sz = record_size(kind);
      if (cnt * sz <= kSmallRunF4) {
        if (!wave_idle) {
          if (logG) hit_records_strided<IMG>(a.blob + off, cblob, kind, cnt, 0, off, j, logG, c, h);
          else hit_records<IMG, 4, 4, true, BADOUEL, false>(cblob + off, cblob, kind, cnt, off, c, fast, L.rng, h);
        }
        continue;
      }
      const int per_tile = kTileF4 / sz;
      for (int first = 0; first < cnt; first += per_tile) {
        const int n = min(per_tile, cnt - first);
        const int nf4 = n * sz, base = off + first * sz;
        for (int i = threadIdx.x; i < nf4; i += kBlock) tile[i] = a.blob[base + i];
        __syncthreads();
        if (!wave_idle) {
          if (logG) hit_records_strided<IMG, false>((lds_f4p)tile, cblob, kind, n, 0, base, j, logG, c, h);
          else hit_records<IMG, 4, 4, false, BADOUEL, false>((lds_f4p)tile, cblob, kind, n, base, c, fast, L.rng, h);
        }
        __syncthreads();
      }
    }
    if (logG) coop_merge_handback<IMG>(h, live_mask, L.live, logG);
    lane_shade<UV, FAST>(L, a, h, a.blob, a.mats);
  }
}

// The reference's single-task executor (render.hpp:113-122, USE_SINGLE_TASK): one default-seeded RNG for the whole frame,
// pixels x-outer / y-inner.  One sequential chain by definition, so one wave runs it (launch <<<1, 64>>>, every lane the same);
// u,v are tracked through the scan as the reference's temp_rec does (no per-scene kernel choice for a parity-only mode).
__global__ __launch_bounds__(64) void render_single_stream_kernel(KArgs a) {
  // all 64 lanes run the same chain redundantly (the traversal's wave-level exchanges need a full wave); lane 0 stores
  const bool writer = threadIdx.x == 0;
  const cst_f4p cblob = (cst_f4p)a.blob;
  uint32_t rng = 2463534242u; // xorshift.hpp:18
  for (int x = 0; x != a.width; ++x)
    for (int y = 0; y != a.height; ++y) {
      V3 acc = mk(0.0f, 0.0f, 0.0f);
      for (int s = 0; s < a.samples; ++s) {
        Ray ray = camera_ray(a.cam, x, y, a.width, a.height, a.inv_w, a.inv_h, rng, a.pinhole != 0);
        V3 att = mk(1.0f, 1.0f, 1.0f), out = mk(0.0f, 0.0f, 0.0f); // depth exhausted: black (render.hpp:91)
        for (int b = 0; b < a.depth; ++b) {
          RayCtx c = make_ctx(ray, a.fast_ok != 0);
          HitState h;
          hit_world<true, true>(a.blob, cblob, a.n_runs, c, c.reg, rng, h);
          if (h.hit < 0) { out = sky_color(ray, att); break; }
          const Rec rec = resolve_hit(a.blob, h.hit, ray, h.closest);
          auto uv = [&](float& u, float& v) { u = h.u; v = h.v; };
          if (!shade(a.mats, a.atlas, rec, uv, ray, att, rng, out)) break;
          out = mk(0.0f, 0.0f, 0.0f);
        }
        acc = acc + out;
      }
      const V3 mean = acc / (float)a.samples;
      float* px = a.fb + ((long long)y * a.width + x) * 3;
      if (writer) { px[0] = mean.x; px[1] = mean.y; px[2] = mean.z; }
    }
}

#include "pt_binned.hpp"

// ---- longest-processing-time-first tile order -------------------------------------------------------------
// A pixel cannot be split (one sequential RNG stream), so the frame's makespan is bounded below by its heaviest
// tile, and a heavy tile picked up LAST adds its whole duration to the tail (measured on the 496-hittable scene:
// mean 2.4 of 5 resident waves per SIMD over the launch).  A probe pass renders the first few samples of every
// pixel and counts rays per tile (same seeds; since round 5 the frame launch resumes from those samples — KArgs.resume_rng); this kernel buckets the tiles
// into 32 classes of ratio 2^(1/4) below the maximum and emits them heaviest class first.  The order only decides
// WHEN a pixel is rendered, never its value, so the (atomic, run-to-run varying) order inside a class is harmless.
// A tile's estimate is its heaviest pixel over a few samples — a noisy number for exactly the pixels that matter (an isolated glass or smoke
// pixel takes 3 rays or 50 per sample), and the two ways of being wrong are not alike: a tile that starts too early costs nothing, a heavy
// tile that starts late is the frame's tail.  So, for whole frames, a tile is ranked by the largest estimate among itself and its eight
// neighbours (a dilation of the cost map): heavy pixels lie along silhouettes that run through neighbouring tiles.  What it removes: one
// launch in five of the 496-hittable 1080p frame took 407 instead of 376 ms — every workgroup resident from the first microsecond, ONE of them
// 40 ms longer than the rest (tools/block_residency.py) — because a tile whose estimate had come out too light sits at a run-to-run varying
// place of its class, and sometimes that place is among the last dequeues.  1080p x 1024 spp 383 - 388 (373 ... 409) -> 375.2 +- 1.5 ms over
// 40 launches, x 256 spp 108 -> 102, 4K x 256 spp 332 (323 ... 352) -> 310 +- 1; neighbours weighted 0.25 / 0.5 / 0.75 / 1.5 or two rings of
// them bring the slow launches back in part (a neighbour ranked above the heavy tile itself, or too many tiles in the top classes);
// averaging a pixel with its neighbours instead — nine times the samples — made it WORSE (404 ms: the heavy pixels are isolated ones).
// profiles/r05_launch_spread_smoke.txt.
__global__ __launch_bounds__(256) void tile_dilate_kernel(const unsigned int* __restrict__ cost, int tiles_x, int tiles_y, unsigned int* __restrict__ out) {
  const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
  if (t >= tiles_x * tiles_y) return;
  const int tx = t % tiles_x, ty = t / tiles_x;
  unsigned int m = 0;
  for (int dy = -1; dy <= 1; dy++)
    for (int dx = -1; dx <= 1; dx++) {
      const int x = tx + dx, y = ty + dy;
      if (x >= 0 && x < tiles_x && y >= 0 && y < tiles_y) m = max(m, cost[y * tiles_x + x]);
    }
  out[t] = m;
}

constexpr int kLptClasses = 32;
__global__ __launch_bounds__(1024) void lpt_order_kernel(const unsigned int* __restrict__ cost, int n, int* __restrict__ order,
                                                        int n_waves, float trav_cost, float fixed_cost, float s_chain, int forced_logG,
                                                        int* __restrict__ n_split) {
  __shared__ float s_cost_sum[kLptClasses];
  __shared__ unsigned int s_max;
  __shared__ float s_sum;
  __shared__ unsigned int s_count[kLptClasses], s_cursor[kLptClasses];
  if (threadIdx.x == 0) { s_max = 1; s_sum = 0.0f; }
  if (threadIdx.x < kLptClasses) { s_count[threadIdx.x] = 0; s_cost_sum[threadIdx.x] = 0.0f; }
  __syncthreads();
  unsigned int m = 0;
  float sum = 0.0f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { m = max(m, cost[i]); sum += (float)cost[i]; }
  atomicMax(&s_max, m);
  atomicAdd(&s_sum, sum);
  __syncthreads();
  const float mx = (float)s_max;
  auto cls = [&](unsigned int c) {
    if (c == 0) return kLptClasses - 1;
    int k = (int)(4.0f * __log2f(mx / (float)c)); // 0 = within 2^(1/4) of the heaviest tile
    return min(max(k, 0), kLptClasses - 1);
  };
  for (int i = threadIdx.x; i < n; i += blockDim.x) {
    const int k = cls(cost[i]);
    atomicAdd(&s_count[k], 1u);
    atomicAdd(&s_cost_sum[k], (float)cost[i]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    // How many of the heaviest tiles to render G lanes per pixel (lane_acquire), and G itself.  A tile rendered whole is one
    // chain of length ~ its cost; wide, its pixels are chains `speedup` times shorter but cost 1/`eff` as much lane time
    // (camera and shading are computed redundantly by the G lanes, the merge is extra).  Per iteration: traversal T
    // (splittable) + S(G) (not; grows with the merge stages):  speedup = (T + S1) / (T/G + S),  eff = (T + S1) / (T + G*S).
    // With the tiles handed out heaviest first the makespan is about
    //     max( (cost kept whole + cost wide / eff) / n_waves , heaviest tile kept whole , heaviest tile / speedup )
    // evaluated for every group size and every prefix of classes; the best pair wins (often the empty prefix).  Small
    // frames and the shards of a multi-GPU job are chain-bound and get large groups; big frames get small ones or none.
    float best = 3.4e38f;
    int split = 0, best_logG = forced_logG > 0 ? forced_logG : 3;
    const float nw = (float)max(n_waves, 1);
    unsigned int acc = 0;
    for (int k = 0; k < kLptClasses; k++) { s_cursor[k] = acc; acc += s_count[k]; }
    for (int logG = (forced_logG > 0 ? forced_logG : 1); logG <= (forced_logG > 0 ? forced_logG : 6); logG++) {
      // Per iteration ~2 400 instruction slots are not split (camera, shading, run headers, merge stages — every lane of a
      // group runs them redundantly) against the list scan's T: lane time of a wide tile grows as (T + G*2400) / (T + 2400),
      // its chain shortens as (T + 2400) / (T/G + 2400).  Refit in round 2 on the 496-hittable scene after the scan got
      // cheaper (tools/model_sweep.py; the round-1 fit charged only 420 slots for the lane time and over-split shards:
      // shard 0/8 of the 1080p frame 113 -> 102 ms, 4K x 512 spp 365 -> 348 ms, whole frames unchanged).
      const float G = (float)(1 << logG);
      const float eff = (trav_cost + fixed_cost) / (trav_cost + G * fixed_cost);
      const float speedup = (trav_cost + s_chain) / (trav_cost / G + s_chain);
      float split_cost = 0.0f;
      unsigned int tiles = 0;
      for (int k = 0; k <= kLptClasses; k++) { // k = number of leading classes that are wide
        const float next_whole = k < kLptClasses ? mx * exp2f(-0.25f * (float)k) : 0.0f; // upper bound of class k
        const float balanced = ((s_sum - split_cost) + split_cost / eff) / nw;
        const float span = fmaxf(balanced, fmaxf(next_whole, k > 0 ? mx / speedup : 0.0f));
        if (span < best * 0.98f) { best = span; split = (int)tiles; best_logG = logG; } // needs a clear win to change
        if (k < kLptClasses) { tiles += s_count[k]; split_cost += s_cost_sum[k]; }
      }
    }
    if (n_split) n_split[1] = best_logG;
    if (n_split) *n_split = split;
  }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += blockDim.x) order[atomicAdd(&s_cursor[cls(cost[i])], 1u)] = i;
}

// ---- probes ---------------------------------------------------------------------------------
template <bool IMG, int WALK = 1>
__global__ void bounce_kernel(const f4* __restrict__ blob, int n_runs, const f4* __restrict__ mats, const f4* __restrict__ pool,
                              const uint8_t* __restrict__ atlas, const PtBounceIn* __restrict__ in,
                              PtBounceOut* __restrict__ outp, int n, int fast_ok) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  // every lane of the wave stays active (the tail repeats the last record and does not store): the traversal exchanges data
  // between lanes (grid walk split phase: ds_bpermute, DPP), and a lane that has exited reads as zero there
  const bool valid_lane = k < n;
  if (!valid_lane) k = n - 1;
  PtBounceIn I = in[k];
  Ray ray;
  ray.o = mk(I.origin[0], I.origin[1], I.origin[2]);
  ray.d = mk(I.dir[0], I.dir[1], I.dir[2]);
  ray.tm = I.time;
  uint32_t rng = I.rng_state;
  V3 att = mk(I.attenuation[0], I.attenuation[1], I.attenuation[2]);
  PtBounceOut O;
  memset(&O, 0, sizeof O);
  RayCtx c = make_ctx(ray, fast_ok != 0);
  HitState h;
  hit_world<IMG, true, WALK, true>(blob, (cst_f4p)blob, n_runs, c, wave_all_regular(c, true), rng, h, pool); // every culling structure the scene has (WALK: which sphere-grid walk)
  const float closest = h.closest, hu = h.u, hv = h.v;
  const int hit = h.hit;
  if (hit < 0) {
    V3 c = sky_color(ray, att);
    O.status = PT_BOUNCE_MISS; O.hittable = -1; O.material = -1;
    O.color[0] = c.x; O.color[1] = c.y; O.color[2] = c.z;
  } else {
    Rec rec = resolve_hit(blob, hit, ray, closest);
    O.hittable = rec.hittable; O.material = rec.mat; O.front_face = rec.front_face ? 1 : 0;
    O.t = closest;
    O.p[0] = rec.p.x; O.p[1] = rec.p.y; O.p[2] = rec.p.z;
    O.normal[0] = rec.normal.x; O.normal[1] = rec.normal.y; O.normal[2] = rec.normal.z;
    float pu = hu, pv = hv;
    if (!IMG) winner_uv(blob, hit, ray, closest, rec, pu, pv);
    O.u = pu; O.v = pv;
    V3 out = mk(0.0f, 0.0f, 0.0f);
    auto uv = [&](float& u, float& v) { u = pu; v = pv; };
    if (shade(mats, atlas, rec, uv, ray, att, rng, out)) {
      O.status = PT_BOUNCE_SCATTERED;
      O.color[0] = att.x; O.color[1] = att.y; O.color[2] = att.z;
      O.sc_origin[0] = ray.o.x; O.sc_origin[1] = ray.o.y; O.sc_origin[2] = ray.o.z;
      O.sc_dir[0] = ray.d.x; O.sc_dir[1] = ray.d.y; O.sc_dir[2] = ray.d.z;
      O.sc_time = ray.tm;
    } else {
      O.status = PT_BOUNCE_ABSORBED;
      O.color[0] = out.x; O.color[1] = out.y; O.color[2] = out.z;
    }
  }
  O.rng_state = rng;
  if (valid_lane) outp[k] = O;
}

__global__ void camera_rays_kernel(Cam cam, int width, int height, const int* __restrict__ xy,
                                   const uint32_t* __restrict__ rng_in, PtCameraRay* __restrict__ out, int n) {
  int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  uint32_t rng = rng_in[k];
  Ray r = camera_ray(cam, xy[2 * k], xy[2 * k + 1], width, height, 1.0f / (float)width, 1.0f / (float)height, rng, cam_is_pinhole(cam));
  PtCameraRay o;
  o.origin[0] = r.o.x; o.origin[1] = r.o.y; o.origin[2] = r.o.z;
  o.dir[0] = r.d.x; o.dir[1] = r.d.y; o.dir[2] = r.d.z;
  o.time = r.tm; o.rng_state = rng;
  out[k] = o;
}

__global__ void math_kernel(int op, const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                            long long n) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float x = a[i], y = b ? b[i] : 0.0f, r;
  switch (op) {
    case 0: r = ptm::sinf_(x); break;
    case 1: r = ptm::cosf_(x); break;
    case 2: r = ptm::logf_(x); break;
    case 3: r = ptm::pow5f_(x); break;
    case 4: r = ptm::atan2f_(x, y); break;
    case 5: r = ptm::asinf_(x); break;
    case 6: r = ptm::fmod1f_(x); break;
    case 7: r = sqrt_rn(x); break;
    case 9: { float yy = 1.0f / y; r = div_exact(x, y, yy, x * yy); break; } // the shared-reciprocal quotient (|q| >= 2^-60)
    case 10: r = rcp_rn_guarded(x); break; // RN(1/a) for 2^-40 <= |a| <= 2^40 (pt_device.hpp: make_ctx)
    case 11: r = sqrt_rn_unit(x); break;   // correctly rounded sqrt for x = 0 or 2^-60 <= x <= 4
    case 12: r = sky_unit_y(x, y, true); break; // unit_vector(d).y = x / sqrt(y) for a regular ray (y = d.d): the sky's shortcut
    case 13: r = checker_sines_negative(x, y, 1.0f) ? 1.0f : 0.0f; break; // texture.hpp:43-45 on (a, b, 1): the checker's sign-only form
    case 14: { float c_; ptm::sincosf_(x, r, c_); break; }  // sin through the fused form
    case 15: { float s_; ptm::sincosf_(x, s_, r); break; }  // cos through the fused form
    default: r = x / y; break;
  }
  out[i] = r;
}

// pt_debug_sphere_texel: the texel an image texture on a sphere selects for the unit normal n — what texture_value takes (the fast
// path where it is unambiguous, the reference's chain otherwise: out_ij), the reference's chain alone (exact_ij), and which one it was.
__global__ void sphere_texel_kernel(const float* __restrict__ nxyz, long long n, float freq, uint32_t w, uint32_t h, int32_t* __restrict__ out_ij,
                                    int32_t* __restrict__ exact_ij, uint8_t* __restrict__ fast, float* __restrict__ uv4) {
  long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const V3 nn = mk(nxyz[3 * k], nxyz[3 * k + 1], nxyz[3 * k + 2]);
  float u, v;
  mercator(nn, u, v);
  const uint32_t ei = texel_index(ptm::fmod1f_(u * freq) * (float)(w - 1), w - 1);
  const uint32_t ej = texel_index((1.0f - ptm::fmod1f_(v * freq)) * (float)(h - 1), h - 1);
  uint32_t i = 0, j = 0;
  const bool took = sphere_texel_fast(nn, freq, w, h, i, j);
  out_ij[2 * k] = (int32_t)(took ? i : ei); out_ij[2 * k + 1] = (int32_t)(took ? j : ej);
  exact_ij[2 * k] = (int32_t)ei; exact_ij[2 * k + 1] = (int32_t)ej;
  fast[k] = took ? 1 : 0;
  if (uv4) { // (u, v) of the reference's chain and of the binary32 approximations: the deviation the short form's margin E_uv covers
    float uf, vf;
    sphere_uv_fast(nn, uf, vf);
    uv4[4 * k] = u; uv4[4 * k + 1] = v; uv4[4 * k + 2] = uf; uv4[4 * k + 3] = vf;
  }
}

// PT_FLAG_FAST_RNG: framebuffer = (chunk plane 0 + plane 1 + ... in order) / samples — a fixed order, so the mode is
// deterministic run to run (and bit-comparable with the oracle's restatement of it).
__global__ void fast_reduce_kernel(const float* __restrict__ partial, float* __restrict__ fb, long long n, long long stride,
                                   int chunks, float samples) {
  long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float sum = 0.0f;
  for (int c = 0; c < chunks; c++) sum = sum + partial[(long long)c * stride + i];
  fb[i] = sum / samples;
}

// Root-side un-interleave of the gathered shard tiles -> [y][x][rgb].
__global__ void unshard_kernel(const float* __restrict__ gathered, float* __restrict__ fb, int width, int height,
                               int tiles_x, int shard_count, int tiles_per_shard) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= width) return;
  int g = (y / PT_TILE) * tiles_x + (x / PT_TILE);
  long long src = (((long long)(g % shard_count) * tiles_per_shard + g / shard_count) * PT_TILE_PIXELS +
                   (y % PT_TILE) * PT_TILE + (x % PT_TILE)) * 3;
  long long dst = ((long long)y * width + x) * 3;
  fb[dst] = gathered[src]; fb[dst + 1] = gathered[src + 1]; fb[dst + 2] = gathered[src + 2];
}

// main.cpp:33-59 — sqrt gamma, clamp [0,0.999], *256, truncate, vertical flip.
__global__ void tonemap_kernel(const float* __restrict__ fb, uint8_t* __restrict__ rgb8, int width, int height) {
  int x = blockIdx.x * blockDim.x + threadIdx.x, row = blockIdx.y; // row 0 = top
  if (x >= width) return;
  int j = height - 1 - row;
  for (int ch = 0; ch < 3; ch++) {
    float s = sqrt_rn(fb[((long long)j * width + x) * 3 + ch]);
    float cl = (s < 0.0f) ? 0.0f : (0.999f < s) ? 0.999f : s; // std::clamp
    float sc = 256.0f * cl;
    int v = (sc == sc) ? (int)sc : 0; // int(NaN) is UB in the reference; defined as 0
    rgb8[((long long)row * width + x) * 3 + ch] = (uint8_t)v;
  }
}

thread_local std::string g_last_error;

// ---- tuning: include/pt_render.h PtTuning.  Resolved ONCE per scene (pt_scene_create / pt_debug_flatten), never on the launch path.
// A NULL PtTuning means the library's defaults with the PT_* environment variables applied on top (the override channel of tools/ and of
// A/B runs); an explicit struct is taken as it is and the environment is not consulted.
static void tuning_defaults(PtTuning& t) {
  std::memset(&t, 0, sizeof t);
  t.struct_size = (int32_t)sizeof(PtTuning);
}
static void tuning_env(PtTuning& t) {
  tuning_defaults(t);
  auto has = [](const char* n) { return std::getenv(n) != nullptr; };
  if (has("PT_NO_GRID")) t.sphere_grid = -1;
  if (const char* e = std::getenv("PT_GRID_M")) t.grid_margin = (float)std::atof(e);
  if (const char* e = std::getenv("PT_GRID_CELL")) t.grid_cell = (float)std::atof(e);
  t.slab_pools = has("PT_NO_BOXCULL") ? -1 : has("PT_POOL_ALWAYS") ? 1 : 0;
  if (has("PT_NO_TRICULL")) t.tri_pool = -1;
  if (has("PT_TRICULL")) t.tri_min_run = 256;
  if (const char* e = std::getenv("PT_TRI_M")) t.tri_M = (float)std::atof(e);
  if (has("PT_TRI_BINNED")) t.tri_binned = 1;
  if (has("PT_NO_TRI_CACHE")) t.tri_cache = -1;
  if (has("PT_NO_SPHERE_MERGE")) t.sphere_merge = -1;
  if (const char* e = std::getenv("PT_TRI_RES")) std::sscanf(e, "%d,%d,%d", &t.tri_res[0], &t.tri_res[1], &t.tri_res[2]);
  if (const char* e = std::getenv("PT_TRI_RHO")) std::sscanf(e, "%f,%f,%f", &t.tri_rho[0], &t.tri_rho[1], &t.tri_rho2);
  if (const char* e = std::getenv("PT_TRI_BUDGET_MB")) t.tri_budget_mb = std::max(1, std::atoi(e));
  if (const char* e = std::getenv("PT_TRI_CELL")) t.tri_cell = (float)std::atof(e);
  if (const char* e = std::getenv("PT_TRI_MIN")) t.tri_min_run = std::max(1, std::atoi(e));
  if (has("PT_NO_MATSPEC")) t.generic_materials = 1;
  if (const char* e = std::getenv("PT_BLOCKS_PER_CU")) t.blocks_per_cu = std::max(1, std::atoi(e));
  if (has("PT_NO_COLD_LDS")) t.cold_state = -1;
  if (const char* e = std::getenv("PT_WIDE_LOGG")) t.wide_log2_group = std::min(6, std::max(1, std::atoi(e)));
  if (const char* e = std::getenv("PT_SPLIT_TILES")) { t.split_tiles_mode = 1; t.split_tiles = std::atoi(e); }
  if (const char* e = std::getenv("PT_LPT_MAX")) t.lpt_by_max = std::atoi(e) != 0 ? 1 : -1;
  if (const char* e = std::getenv("PT_PROBE_SPP_MAX")) t.probe_spp_max = std::max(1, std::atoi(e));
  if (has("PT_NO_PROBE_RESUME")) t.probe_resume = -1;
  if (has("PT_NO_CHAIN_PRIO")) t.chain_priority = -1;
  if (const char* e = std::getenv("PT_GRID_MIN_TILES")) t.grid_min_tiles = std::max(0, std::atoi(e));
  if (const char* e = std::getenv("PT_MODEL_FIXED")) t.model_fixed = (float)std::atof(e);
  if (const char* e = std::getenv("PT_MODEL_CHAIN")) t.model_chain = (float)std::atof(e);
  if (const char* e = std::getenv("PT_SCATTER_LOG")) t.scatter_log = std::min(5, std::max(0, std::atoi(e)));
  t.scatter_mode = has("PT_NO_SCATTER") ? -1 : has("PT_LPT_SCATTER") ? 1 : 0;
  if (const char* e = std::getenv("PT_GRID_WALK")) t.grid_walk = std::atoi(e);
  if (const char* e = std::getenv("PT_LANES_CAP")) t.lanes_cap = std::atoi(e) <= 0 ? -1 : std::min(64, std::atoi(e));
  if (const char* e = std::getenv("PT_HEAVY_TILES")) t.heavy_tiles = std::atoi(e) <= 0 ? -1 : std::atoi(e);
}
// the caller's struct (possibly from an older header: struct_size bytes are valid) or, for NULL, defaults + environment
static int resolve_tuning(const PtTuning* user, PtTuning& t, std::string& err) {
  if (!user) { tuning_env(t); return PT_OK; }
  if (user->struct_size < 8 || user->struct_size > (int32_t)sizeof(PtTuning)) { err = "PtTuning.struct_size is not a size this library knows (use pt_tuning_init)"; return PT_ERR_INVALID_ARG; }
  tuning_defaults(t);
  std::memcpy(&t, user, (size_t)user->struct_size);
  t.struct_size = (int32_t)sizeof(PtTuning);
  return PT_OK;
}

// The flattening pt_scene_create uploads (pt_debug_flatten shows the same blob).
// LDS budget: the culling grid's tables ride in the blob, and only the LDS-resident kernels walk the grid.  A scene whose
// blob exceeds kMaxLdsBlob WITH its grid but fits WITHOUT it (e.g. 1 200 small spheres: 101 KB against 62 KB) is flattened
// without the grid, so that it keeps the resident kernels instead of falling to the streaming kernel with dead tables.
static int flatten_tuned(const PtSceneDesc* desc, const PtTuning& t, ptf::Flat& flat, std::string& err) {
  const int box_cull = t.slab_pools < 0 ? 0 : t.slab_pools > 0 ? 2 : 1;
  ptf::GridTuning tune;
  if (t.grid_margin > 0.0f) tune.m = t.grid_margin;
  if (t.grid_cell > 0.0f) tune.cell = t.grid_cell;
  const bool allow_grid = t.sphere_grid >= 0;
  // Exact culling of long triangle runs (pt_tripool.hpp; pt_device.hpp: tri_pool_scan).  Default: runs of >= 4096 triangles get a
  // pool (BASELINE config 5, 100 k triangles: 68.9 -> 18.6 s per frame, bit-identical).
  bool allow_tri = t.tri_pool >= 0;
  for (int i = 0; desc && desc->hittables && i < desc->n_hittables && allow_tri; i++) // scenes with Badouel-strategy triangles render through the
    if (desc->hittables[i].kind == PT_HIT_TRIANGLE && desc->hittables[i].strategy == PT_TRI_BADOUEL) allow_tri = false; // round-2 kernels: no dead tables
  ptf::TriPoolTuning tri;
  if (t.tri_min_run > 0) tri.min_run = t.tri_min_run;
  if (t.tri_M > 0.0f) tri.M = t.tri_M;
  for (int k = 0; k < 3; k++) if (t.tri_res[k] > 0) tri.dm_res[k] = t.tri_res[k];
  if (t.tri_rho[0] != 0.0f) tri.dm_rho[0] = t.tri_rho[0]; // (< 0: no such map)
  if (t.tri_rho[1] != 0.0f) tri.dm_rho[1] = t.tri_rho[1];
  if (t.tri_rho2 != 0.0f) tri.dm_rho[2] = t.tri_rho2;
  if (t.tri_budget_mb > 0) tri.dm_budget = (long long)t.tri_budget_mb * (1 << 18); // MiB -> 4-byte entries
  if (t.tri_cell > 0.0f) tri.cell = t.tri_cell;
  if (const char* e = std::getenv("PT_TRI_GRID_BUDGET")) tri.grid_budget = (float)std::atof(e); // (experiments only: not a PtTuning field)
  const bool merge = t.sphere_merge >= 0;
  int rc = ptf::flatten(desc, flat, err, allow_grid, box_cull, tune, allow_tri, tri, merge);
  if (rc) return rc;
  if (flat.grid_spheres > 0 && flat.blob.size() * 16 > kMaxLdsBlob) {
    ptf::Flat plain;
    std::string err2;
    if (flat.tri_pooled == 0 && ptf::flatten(desc, plain, err2, false, box_cull, tune, allow_tri, tri, merge) == PT_OK && plain.blob.size() * 16 <= kMaxLdsBlob) flat = std::move(plain);
  }
  return PT_OK;
}

int fail(int code, const std::string& msg) {
  g_last_error = msg;
  return code;
}

#define PT_HIP(expr)                                                                             \
  do {                                                                                           \
    hipError_t e_ = (expr);                                                                      \
    if (e_ != hipSuccess) return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)

// The launch-side knobs of a scene's PtTuning, resolved once in pt_scene_create — never on the launch path.
struct Knobs {
  int blocks_per_cu = 0;   // cap on resident workgroups per CU (0 = none)
  bool no_cold_lds = false;
  int wide_logG = 0;       // forced log2 group size of the wide phase (0 = the model picks)
  bool has_split_tiles = false;
  int split_tiles = 0;     // fixed number of tiles through the wide phase (< 0: all)
  int lpt_max = -1, probe_spp_max = 16; // order tiles by their heaviest pixel (-1: by kernel family); probe depth cap
  int grid_min_tiles = kGridMinTiles;    // frames (shards) of fewer tiles keep the cooperative kernels and the lists
  float model_fixed = 2400.0f, model_chain = 2400.0f; // constants of the makespan model (lpt_order_kernel)
  int scatter_log = 0;     // log2 of the pixels of one tile that a wave takes together (0: every lane a pixel of another tile)
  bool lpt_with_scatter = false; // the cost probe + heaviest-first order also for scattered (triangle-pool) renders
  int grid_block = kGridBlock; // PT_GRID_BLOCK (build-time experiment: workgroup size of the grid kernels)
  bool no_scatter = false; // triangle-pool kernels hand out whole tiles' pixels to a wave again (lane_acquire)
  bool generic_materials = false;
  int grid_walk = 0;       // PtTuning.grid_walk: 0 the launcher's rule, 1 the wave-synchronous walk, 2 the queued walk
  int lanes_cap = 0;       // PtTuning.lanes_cap: grid kernels on small frames (launch): 0 the rule, -1 whole tiles always, n forced
  bool no_tri_cache = false;  // PtTuning.tri_cache = -1: camera rays enumerate their direction-map list like every other ray
  bool no_chain_prio = false; // PtTuning.chain_priority = -1: no issue priorities in the headline family's frame launches
  bool no_resume = false;  // PtTuning.probe_resume = -1: the probe's samples are rendered again by the frame launch
  int heavy_tiles = 0;     // PtTuning.heavy_tiles: tiles at the head of the cost-sorted order that are handed out 16 pixels at a time: 0 the rule, -1 never, n forced
  Knobs() {}
  explicit Knobs(const PtTuning& t) {
    blocks_per_cu = std::max(0, t.blocks_per_cu);
    no_cold_lds = t.cold_state < 0;
    wide_logG = std::min(6, std::max(0, t.wide_log2_group));
    has_split_tiles = t.split_tiles_mode == 1; split_tiles = t.split_tiles;
    lpt_max = t.lpt_by_max > 0 ? 1 : t.lpt_by_max < 0 ? 0 : -1;
    if (t.probe_spp_max > 0) probe_spp_max = t.probe_spp_max;
    if (t.grid_min_tiles > 0) grid_min_tiles = t.grid_min_tiles;
    if (t.model_fixed > 0.0f) model_fixed = t.model_fixed;
    if (t.model_chain > 0.0f) model_chain = t.model_chain;
    scatter_log = std::min(5, std::max(0, t.scatter_log));
    no_scatter = t.scatter_mode < 0; lpt_with_scatter = t.scatter_mode > 0;
    generic_materials = t.generic_materials != 0;
    lanes_cap = t.lanes_cap < 0 ? -1 : std::min(64, t.lanes_cap);
    grid_walk = (t.grid_walk == 1 || t.grid_walk == 2) ? t.grid_walk : 0;
    heavy_tiles = t.heavy_tiles < 0 ? -1 : t.heavy_tiles;
    no_resume = t.probe_resume < 0;
    no_chain_prio = t.chain_priority < 0;
    no_tri_cache = t.tri_cache < 0;
  }
};

struct EventPair { // RAII: no leak on an early return
  hipEvent_t e0 = nullptr, e1 = nullptr;
  ~EventPair() { if (e0) (void)hipEventDestroy(e0); if (e1) (void)hipEventDestroy(e1); }
};

template <typename T>
struct DevBuf {
  T* p = nullptr;
  ~DevBuf() { if (p) (void)hipFree(p); }
  hipError_t alloc(size_t n) { return hipMalloc((void**)&p, std::max<size_t>(n, 1) * sizeof(T)); }
};

} // namespace

struct PtScene {
  f4* blob = nullptr;
  f4* mats = nullptr;
  f4* pool = nullptr;   // tables of the triangle pools (a buffer of their own: up to gigabytes)
  size_t pool_bytes = 0;
  uint8_t* atlas = nullptr;
  int n_runs = 0, blob_f4 = 0, mats_f4 = 0;
  bool has_image = false;
  bool track_uv = false; // an image texture sits on a triangle or a medium: the stale u,v such hits inherit must be tracked
  bool fast_ok = false;
  bool coop_ok = false;
  bool has_badouel = false; // Badouel-strategy triangles: the scalar-cache / streaming kernels compiled with their loop
  int coop_prefix = 0;
  int n_hittables = 0;
  mutable bool last_had_wide_phase = false;
  mutable int last_launch[4] = {0, 0, 0, 0}; // what the launcher decided for the last frame launch: workgroups, lanes_cap, heavy_pixels, queued walk (pt_debug_last_launch)
  mutable int nsplit_override = 0; // PT_SPLIT_TILES tuning knob (host copy must outlive the async upload)
  float traversal_cost = 0.0f; // estimated VALU instructions of one ray's scan of the list (sphere runs through their lists)
  int grid_spheres = 0;        // spheres that sit in a culling grid (the resident non-cooperative kernels walk it)
  int tri_pooled = 0;          // triangles that sit in a triangle pool (the TRIPOOL kernels query it)
  bool rectbox_only = false;   // every hittable is a rect or a box (kernels compiled with MATS_RECTBOX_ONLY: resolve_hit)
  bool mats_simple = false;    // every material is lambertian or lightsource over a solid texture (kernels compiled with MATS_LAMB_LIGHT_SOLID)
  size_t blob_bytes = 0, atlas_bytes = 0;
  int num_cus = 256;
  size_t lds_per_block = 64 * 1024; // hipDeviceProp_t::sharedMemPerBlock (gfx950: 160 KB; the Makefile's ARCH=gfx942: 64 KB)
  mutable unsigned int* ws_cost = nullptr; // LPT workspace: per-tile ray counts of the probe pass
  mutable unsigned int* ws_rank = nullptr; // per local tile: the cost a tile is ranked by (tile_dilate_kernel)
  mutable unsigned int* ws_rng = nullptr;  // per local pixel: the generator's state after the probe's samples (KArgs.resume_rng)
  mutable int* ws_order = nullptr;         //                cost-sorted tile order
  mutable int ws_tiles = 0;
  int* ws_nsplit = nullptr;                //                number of leading tiles to split (device scalar)
  mutable float* ws_partial = nullptr;     // PT_FLAG_FAST_RNG: per-chunk partial sums (grow-only)
  mutable size_t ws_partial_floats = 0;
  // the binned triangle-pool renderer (pt_binned.hpp): the pooled run, the key space of its direction maps, and a grow-only workspace
  int bin_run = -1, bin_hdr = 0, bin_goff = 0, bin_keys = 0, bin_full_slices = 1, bin_base1 = 0;
  bool binned = false;          // this scene's parity-mode renders go through it
  mutable unsigned int* ws_tricache = nullptr; // the camera rays' candidate cache: one line per resident lane of the triangle-pool kernels
  mutable size_t ws_tricache_lanes = 0;
  mutable void* ws_bin = nullptr; // per-pixel state, requests and slots for ws_bin_pixels local pixels; the sort's tables
  mutable size_t ws_bin_pixels = 0;
  mutable int last_generations = 0; // generations of the last binned render (pt_debug_last_launch)
  unsigned int* queues = nullptr; // ring of per-launch pixel-queue counters
  mutable unsigned int next_queue = 0;
  mutable hipEvent_t ring_done[kQueueRing] = {}; // recorded behind the launch that uses a slot: a wrapped ring waits for it
  int device = 0;
  // Scheduling state above marked `mutable` (queue ring cursor, LPT workspace, last-launch info) changes per launch although
  // the scene DATA is immutable: launches on one scene from several host threads serialise their ENQUEUE on this mutex.
  // The per-scene workspaces (ws_cost / ws_order / ws_rng / ws_nsplit / ws_partial) are shared by every launch on the scene, so
  // renders on ONE scene must also be stream-ordered (include/pt_render.h): callers that render one scene from several
  // streams create one PtScene per stream (path_tracer_amd/render.py keys its cache by (device, stream)).
  mutable std::mutex sched;
  mutable std::map<const void*, int> occupancy; // resident workgroups per CU, per kernel variant (queried once)
  Knobs knobs;                                  // the scene's PtTuning (launch side), as resolved when the scene was created
};

extern "C" {

int pt_abi_version(void) { return PT_ABI_VERSION; }

const char* pt_error_string(int code) {
  switch (code) {
    case PT_OK: return "ok";
    case PT_ERR_INVALID_ARG: return "invalid argument";
    case PT_ERR_BAD_SCENE: return "malformed scene tables";
    case PT_ERR_HIP: return "HIP runtime error";
    case PT_ERR_NO_DEVICE: return "no HIP device";
    case PT_ERR_TOO_LARGE: return "scene too large";
    default: return "unknown error";
  }
}

const char* pt_last_error(void) { return g_last_error.c_str(); }

// camera.hpp:67-87 (host arithmetic, binary32, no contraction)
int pt_camera_init(PtCamera* cam, const float look_from[3], const float look_at[3], const float vup[3],
                   float vfov_deg, float aspect_ratio, float aperture, float focus_dist, float time0, float time1) {
  if (!cam || !look_from || !look_at || !vup) return fail(PT_ERR_INVALID_ARG, "pt_camera_init: NULL argument");
  struct H3 { float x, y, z; };
  auto sub = [](H3 a, H3 b) { return H3{a.x - b.x, a.y - b.y, a.z - b.z}; };
  auto scale = [](float s, H3 a) { return H3{s * a.x, s * a.y, s * a.z}; };
  auto divs = [](H3 a, float s) { return H3{a.x / s, a.y / s, a.z / s}; };
  auto dot3 = [](H3 a, H3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; };
  auto cross3 = [](H3 a, H3 b) { return H3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; };
  auto unit = [&](H3 a) { return divs(a, std::sqrt(dot3(a, a))); };
  const float pi = 3.1415926535897932385f;
  H3 origin{look_from[0], look_from[1], look_from[2]}, at{look_at[0], look_at[1], look_at[2]}, up{vup[0], vup[1], vup[2]};
  float theta = vfov_deg * pi / 180.0f; // rtweekend.hpp:31
  float h = std::tan(theta / 2.0f);
  float viewport_height = 2.0f * h;
  float viewport_width = aspect_ratio * viewport_height;
  H3 w = unit(sub(origin, at));
  H3 u = unit(cross3(up, w));
  H3 v = cross3(w, u);
  H3 horizontal = scale(focus_dist * viewport_width, u);
  H3 vertical = scale(focus_dist * viewport_height, v);
  H3 llc = sub(sub(sub(origin, divs(horizontal, 2.0f)), divs(vertical, 2.0f)), scale(focus_dist, w));
  auto st = [](float* d, H3 a) { d[0] = a.x; d[1] = a.y; d[2] = a.z; };
  st(cam->origin, origin); st(cam->lower_left_corner, llc); st(cam->horizontal, horizontal); st(cam->vertical, vertical);
  st(cam->u, u); st(cam->v, v); st(cam->w, w);
  cam->lens_radius = aperture / 2.0f;
  cam->time0 = time0;
  cam->time1 = time1;
  return PT_OK;
}

// Host-only view of the flattening (no GPU needed): blob_out/mats_out may be NULL to query sizes.
void pt_tuning_init(PtTuning* t) { if (t) tuning_defaults(*t); }
void pt_tuning_from_env(PtTuning* t) { if (t) tuning_env(*t); }

int pt_debug_flatten(const PtSceneDesc* desc, float* blob_out, int64_t blob_cap_f4, int32_t* n_blob_f4,
                     int32_t* n_runs, float* mats_out, int64_t mats_cap_f4, int32_t* flags_out) {
  return pt_debug_flatten_tuned(desc, nullptr, blob_out, blob_cap_f4, n_blob_f4, n_runs, mats_out, mats_cap_f4, flags_out);
}

int pt_debug_flatten_tuned(const PtSceneDesc* desc, const PtTuning* tuning, float* blob_out, int64_t blob_cap_f4, int32_t* n_blob_f4,
                           int32_t* n_runs, float* mats_out, int64_t mats_cap_f4, int32_t* flags_out) {
  ptf::Flat flat;
  std::string err;
  PtTuning t;
  int rc = resolve_tuning(tuning, t, err);
  if (rc) return fail(rc, err);
  rc = flatten_tuned(desc, t, flat, err); // the blob pt_scene_create would upload, knobs and LDS budget included
  if (rc) return fail(rc, err);
  if (n_blob_f4) *n_blob_f4 = (int32_t)flat.blob.size();
  if (n_runs) *n_runs = flat.n_runs;
  if (flags_out) *flags_out = (flat.has_image ? 1 : 0) | (flat.has_medium ? 2 : 0) | (flat.tri_pooled ? 4 : 0);
  if (blob_out) {
    if (blob_cap_f4 < (int64_t)flat.blob.size()) return fail(PT_ERR_INVALID_ARG, "blob buffer too small");
    if (!flat.blob.empty()) std::memcpy(blob_out, flat.blob.data(), flat.blob.size() * 16);
  }
  if (mats_out) {
    if (mats_cap_f4 < (int64_t)flat.mats.size()) return fail(PT_ERR_INVALID_ARG, "material buffer too small");
    if (!flat.mats.empty()) std::memcpy(mats_out, flat.mats.data(), flat.mats.size() * 16);
  }
  return PT_OK;
}

int pt_debug_flatten_pool(const PtSceneDesc* desc, const PtTuning* tuning, float* pool_out, int64_t pool_cap_f4, int64_t* n_pool_f4) {
  ptf::Flat flat;
  std::string err;
  PtTuning t;
  int rc = resolve_tuning(tuning, t, err);
  if (rc) return fail(rc, err);
  rc = flatten_tuned(desc, t, flat, err);
  if (rc) return fail(rc, err);
  if (n_pool_f4) *n_pool_f4 = (int64_t)flat.pool.size_f4;
  if (pool_out) {
    if (pool_cap_f4 < (int64_t)flat.pool.size_f4) return fail(PT_ERR_INVALID_ARG, "pool buffer too small");
    flat.pool.assemble((ptf::F4*)pool_out);
  }
  return PT_OK;
}

int pt_debug_tri_pool(const PtSceneDesc* desc, int32_t out[8]) {
  if (!out) return fail(PT_ERR_INVALID_ARG, "pt_debug_tri_pool: NULL argument");
  ptf::Flat flat;
  std::string err;
  PtTuning t;
  tuning_env(t);
  int rc = flatten_tuned(desc, t, flat, err);
  if (rc) return fail(rc, err);
  out[0] = flat.tri_pooled; out[1] = flat.tri_wide;
  out[2] = (int32_t)(flat.tri_map_entries[0] >> 10); out[3] = (int32_t)((flat.tri_map_entries[1] + flat.tri_map_entries[2]) >> 10);
  out[4] = (flat.tri_map_res[0] << 20) | (flat.tri_map_res[1] << 10) | flat.tri_map_res[2];
  out[5] = (int32_t)(1000.0 * flat.tri_cells_per_triangle); out[6] = (int32_t)flat.blob.size(); out[7] = flat.grid_spheres;
  return PT_OK;
}

int pt_scene_create(const PtSceneDesc* desc, PtScene** out_scene) { return pt_scene_create_tuned(desc, nullptr, out_scene); }

int pt_scene_create_tuned(const PtSceneDesc* desc, const PtTuning* tuning, PtScene** out_scene) {
  if (!out_scene) return fail(PT_ERR_INVALID_ARG, "pt_scene_create: out_scene is NULL");
  *out_scene = nullptr;
  ptf::Flat flat;
  std::string err;
  PtTuning tun;
  int rc = resolve_tuning(tuning, tun, err);
  if (rc) return fail(rc, err);
  const bool timing = std::getenv("PT_TRI_TIMING") != nullptr; // (diagnostics: stage times on stderr, like build_tri_pool's)
  auto t_prev = std::chrono::steady_clock::now();
  auto lap = [&](const char* what) {
    if (!timing) return;
    const auto t = std::chrono::steady_clock::now();
    std::fprintf(stderr, "pt_scene_create: %-27s %.3f s\n", what, std::chrono::duration<double>(t - t_prev).count());
    t_prev = t;
  };
  rc = flatten_tuned(desc, tun, flat, err);
  if (rc) return fail(rc, err);
  lap("flatten (+ culling tables)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible");
  PtScene* s = new PtScene();
  s->knobs = Knobs(tun);
  auto cleanup = [&]() { pt_scene_destroy(s); };
  hipError_t e;
#define PT_TRY(expr) if ((e = (expr)) != hipSuccess) { cleanup(); return fail(PT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e)); }
  PT_TRY(hipGetDevice(&s->device));
  {
    hipDeviceProp_t prop;
    PT_TRY(hipGetDeviceProperties(&prop, s->device));
    s->num_cus = prop.multiProcessorCount;
    s->lds_per_block = prop.sharedMemPerBlock;
  }
  s->n_runs = flat.n_runs;
  s->blob_f4 = (int)flat.blob.size();
  s->has_image = flat.has_image;
  s->fast_ok = flat.fast_ok;
  s->coop_ok = flat.coop_ok && !flat.has_badouel; // the cooperative scans do not carry the Badouel loop
  s->has_badouel = flat.has_badouel;
  s->track_uv = flat.has_image && !flat.coop_ok;
  s->coop_prefix = flat.coop_prefix;
  s->n_hittables = desc->n_hittables;
  for (int i = 0; i < desc->n_hittables; i++) {
    switch (desc->hittables[i].kind) {
      case PT_HIT_SPHERE: s->traversal_cost += 22.0f; break;
      case PT_HIT_TRIANGLE: s->traversal_cost += 35.0f; break;
      case PT_HIT_BOX: s->traversal_cost += 120.0f; break;
      case PT_HIT_CONSTANT_MEDIUM: s->traversal_cost += 300.0f; break;
      default: s->traversal_cost += 20.0f; break;
    }
  }
  s->grid_spheres = flat.grid_spheres;
  s->rectbox_only = desc->n_hittables > 0 && !s->knobs.generic_materials;
#ifdef PT_NO_RECTBOX /* A/B build */
  s->rectbox_only = false;
#endif
  for (int i = 0; i < desc->n_hittables && s->rectbox_only; i++) {
    const int k = desc->hittables[i].kind;
    if (k != PT_HIT_XY_RECT && k != PT_HIT_XZ_RECT && k != PT_HIT_YZ_RECT && k != PT_HIT_BOX) s->rectbox_only = false;
  }
  s->mats_simple = desc->n_materials > 0 && !s->knobs.generic_materials; // PtTuning.generic_materials: A/B knob (generic shading)
  for (int i = 0; i < desc->n_materials && s->mats_simple; i++) {
    const PtMaterial& m = desc->materials[i];
    if ((m.kind != PT_MAT_LAMBERTIAN && m.kind != PT_MAT_LIGHTSOURCE) || desc->textures[m.texture].kind != PT_TEX_SOLID) s->mats_simple = false;
  }
  s->tri_pooled = flat.has_badouel ? 0 : flat.tri_pooled; // (scenes with Badouel-strategy triangles keep the round-2 kernels)
  // The binned renderer serves scenes with ONE pooled run whose stale u, v need no tracking (pt_binned.hpp); PtTuning.tri_binned = 1 (measured: it does not pay yet — docs/EXPERIMENTS.md — so it is opt-in).
  if (s->tri_pooled > 0 && flat.tri_pool_runs == 1 && !s->track_uv && tun.tri_binned > 0) {
    s->binned = true;
    s->bin_run = flat.tri_pool_run; s->bin_hdr = flat.tri_pool_hdr; s->bin_goff = flat.tri_pool_goff;
    s->bin_full_slices = std::min(64, std::max(1, (flat.tri_pool_count + 2047) / 2048));
    long long keys = 2; // ... + "every band record" + "every triangle, exactly" (pt_device.hpp: tri_pool_scan<true>)
    for (int k = 0; k < flat.tri_maps && k < 3; k++) keys += 3ll * flat.tri_map_res[k] * flat.tri_map_res[k];
    s->bin_keys = (int)keys;
    s->bin_base1 = flat.tri_maps > 0 ? 3 * flat.tri_map_res[0] * flat.tri_map_res[0] : 0;
  }
  s->blob_bytes = flat.blob.size() * 16;
  s->mats_f4 = (int)flat.mats.size();
  // one buffer: [blob records][material table] so a kernel can stage both with one contiguous copy
  const size_t blob_bytes = flat.blob.size() * 16, mats_bytes = flat.mats.size() * 16;
  PT_TRY(hipMalloc((void**)&s->blob, std::max<size_t>(blob_bytes + mats_bytes, 16)));
  s->mats = s->blob + flat.blob.size();
  if (blob_bytes) PT_TRY(hipMemcpy(s->blob, flat.blob.data(), blob_bytes, hipMemcpyHostToDevice));
  if (mats_bytes) PT_TRY(hipMemcpy(s->mats, flat.mats.data(), mats_bytes, hipMemcpyHostToDevice));
  if (flat.pool.size_f4) { // the triangle pools' tables: zeroed (spare records between the tables), then table by table from where the builder left them
    s->pool_bytes = (size_t)flat.pool.size_f4 * 16;
    PT_TRY(hipMalloc((void**)&s->pool, s->pool_bytes));
    PT_TRY(hipMemset(s->pool, 0, s->pool_bytes));
    if (timing) { (void)hipDeviceSynchronize(); lap("blob upload, pool malloc + memset"); }
    // (the big tables — hundreds of megabytes of direction-map lists — are pinned in place for their copy: a pageable hipMemcpy
    // staged them at ~3 GB/s, a third of the scene's build time in round 6)
    for (const ptf::PoolSegment& sg : flat.pool.segments) {
      if (sg.dwords.empty()) continue;
      const size_t bytes = sg.dwords.size() * 4;
      const bool pin = bytes >= (8u << 20) && hipHostRegister((void*)sg.dwords.data(), bytes, hipHostRegisterDefault) == hipSuccess;
      e = hipMemcpy(s->pool + sg.at_f4, sg.dwords.data(), bytes, hipMemcpyHostToDevice);
      if (pin) (void)hipHostUnregister((void*)sg.dwords.data());
      if (e != hipSuccess) { cleanup(); return fail(PT_ERR_HIP, std::string("hipMemcpy of a pool table: ") + hipGetErrorString(e)); }
    }
  }
  lap("upload blob + pool");
  size_t atlas_bytes = flat.has_image ? (size_t)desc->atlas_bytes : 0;
  s->atlas_bytes = atlas_bytes;
  PT_TRY(hipMalloc((void**)&s->atlas, std::max<size_t>(atlas_bytes, 16)));
  if (atlas_bytes) PT_TRY(hipMemcpy(s->atlas, desc->atlas, atlas_bytes, hipMemcpyHostToDevice));
  PT_TRY(hipMalloc((void**)&s->queues, 2 * kQueueRing * sizeof(unsigned int)));
  PT_TRY(hipMalloc((void**)&s->ws_nsplit, 2 * sizeof(int))); // [0] tiles through the wide phase, [1] log2 of its group size
#undef PT_TRY
  *out_scene = s;
  return PT_OK;
}

int64_t pt_scene_device_bytes(const PtScene* s) {
  if (!s) return -1;
  return (int64_t)s->blob_bytes + (int64_t)s->mats_f4 * 16 + (int64_t)s->pool_bytes + (int64_t)s->atlas_bytes;
}

#ifndef PT_BUILD_ID
#define PT_BUILD_ID "unknown"
#endif
const char* pt_build_id(void) { return PT_BUILD_ID; }

void pt_scene_destroy(PtScene* s) {
  if (!s) return;
  if (s->blob) (void)hipFree(s->blob);
  if (s->pool) (void)hipFree(s->pool);
  if (s->atlas) (void)hipFree(s->atlas);
  if (s->queues) (void)hipFree(s->queues);
  for (hipEvent_t e : s->ring_done) if (e) (void)hipEventDestroy(e);
  if (s->ws_cost) (void)hipFree(s->ws_cost);
  if (s->ws_order) (void)hipFree(s->ws_order);
  if (s->ws_rng) (void)hipFree(s->ws_rng);
  if (s->ws_rank) (void)hipFree(s->ws_rank);
  if (s->ws_nsplit) (void)hipFree(s->ws_nsplit);
  if (s->ws_partial) (void)hipFree(s->ws_partial);
  if (s->ws_bin) (void)hipFree(s->ws_bin);
  if (s->ws_tricache) (void)hipFree(s->ws_tricache);
  delete s;
}

static int check_params(const PtRenderParams* p) {
  if (!p) return fail(PT_ERR_INVALID_ARG, "render params are NULL");
  if (p->width <= 0 || p->height <= 0 || p->samples <= 0 || p->depth < 0)
    return fail(PT_ERR_INVALID_ARG, "width, height, samples must be > 0 and depth >= 0");
  if (p->shard_count < 1 || p->shard_index < 0 || p->shard_index >= p->shard_count)
    return fail(PT_ERR_INVALID_ARG, "need 0 <= shard_index < shard_count");
  return PT_OK;
}

static int n_tiles_of(const PtRenderParams* p, int* tiles_x) {
  int tx = (p->width + PT_TILE - 1) / PT_TILE, ty = (p->height + PT_TILE - 1) / PT_TILE;
  if (tiles_x) *tiles_x = tx;
  return tx * ty;
}

uint32_t pt_fast_seed(uint32_t pixel, uint32_t chunk) { return fast_seed(pixel, chunk); }

int32_t pt_shard_tiles(const PtRenderParams* p) {
  if (check_params(p)) return -1;
  int n = n_tiles_of(p, nullptr);
  return (n + p->shard_count - 1) / p->shard_count;
}

int64_t pt_framebuffer_floats(const PtRenderParams* p) {
  if (check_params(p)) return -1;
  if (p->shard_count == 1) return (int64_t)p->width * p->height * 3;
  return (int64_t)pt_shard_tiles(p) * PT_TILE_PIXELS * 3;
}

// grow-only per-scene workspaces (LPT cost / order arrays; fast mode's partial sums).  pt_scene_reserve() sizes them ahead of
// time so that pt_render() neither allocates nor frees (hipMalloc / hipFree synchronise the device).
static int reserve_tiles(const PtScene* s, int local_tiles) {
  if (s->ws_tiles >= local_tiles) return PT_OK;
  if (s->ws_cost) (void)hipFree(s->ws_cost);
  if (s->ws_order) (void)hipFree(s->ws_order);
  if (s->ws_rng) (void)hipFree(s->ws_rng);
  if (s->ws_rank) (void)hipFree(s->ws_rank);
  s->ws_cost = nullptr; s->ws_order = nullptr; s->ws_rng = nullptr; s->ws_rank = nullptr; s->ws_tiles = 0;
  PT_HIP(hipMalloc((void**)&s->ws_cost, (size_t)local_tiles * sizeof(unsigned int)));
  PT_HIP(hipMalloc((void**)&s->ws_order, (size_t)local_tiles * sizeof(int)));
  PT_HIP(hipMalloc((void**)&s->ws_rng, (size_t)local_tiles * PT_TILE_PIXELS * sizeof(unsigned int)));
  PT_HIP(hipMalloc((void**)&s->ws_rank, (size_t)local_tiles * sizeof(unsigned int)));
  s->ws_tiles = local_tiles;
  return PT_OK;
}
static int reserve_partial(const PtScene* s, size_t floats) {
  if (s->ws_partial_floats >= floats) return PT_OK;
  if (s->ws_partial) (void)hipFree(s->ws_partial);
  s->ws_partial = nullptr; s->ws_partial_floats = 0;
  PT_HIP(hipMalloc((void**)&s->ws_partial, floats * sizeof(float)));
  s->ws_partial_floats = floats;
  return PT_OK;
}

// The binned triangle-pool renderer (pt_binned.hpp): generations of { step, sort the requests by direction bin, band stage }, until no pixel
// is live.  How many generations a frame takes is known only on the device (the heaviest pixel's ray count), so the host reads the live
// count back every few generations: unlike the persistent kernels' launches this render has returned only when the frame is done.
static int reserve_binned(const PtScene* s, size_t n_pixels, size_t n_keys, size_t& bytes_out);
static int launch_binned(const PtScene* s, KArgs a, const PtRenderParams* p, int local_tiles, hipStream_t st) {
  if (p->depth <= 0) return PT_OK; // every sample black (render.hpp:58,91): the frame is pre-zeroed
  a.n_local_pixels = local_tiles * PT_TILE_PIXELS;
  a.cost = nullptr; a.cost_max = 0; a.resume_rng = nullptr; a.resume_spp = 0; a.order = nullptr; a.n_split = nullptr;
  a.prio_onset = a.prio_t1 = a.prio_t2 = a.prio_t3 = 0;
  a.tile_granular = 1; a.tile_group = 64; a.heavy_pixels = 0; a.heavy_lanes = 64; a.scatter_p = 0; a.scatter_log = 0; a.lanes_cap = 64;
  a.fast_chunks = 0; a.samples_total = p->samples; a.fast_stride = 0; a.queue = nullptr; a.coop_prefix = -1;
  const size_t N = (size_t)a.n_local_pixels, K = (size_t)s->bin_keys;
  size_t bytes = 0;
  if (int rc = reserve_binned(s, N, K, bytes)) return rc;
  // carve the workspace (every array 256-byte aligned)
  char* base = (char*)s->ws_bin;
  size_t at = 0;
  auto take = [&](size_t n) { char* q = base + at; at += (n + 255) & ~(size_t)255; return q; };
  const size_t n_blocks = (K + 1023) / 1024, max_packets = (N / 64 + 1) * (size_t)s->bin_full_slices + std::min(K, N) + 1;
  BinArgs ba;
  ba.k = a;
  ba.A0 = (f4*)take(N * 16); ba.A1 = (f4*)take(N * 16); ba.A2 = (f4*)take(N * 16); ba.A3 = (f4*)take(N * 16);
  ba.A4 = (int4*)take(N * 16); ba.A5 = (f4*)take(N * 16);
  ba.slot = (unsigned long long*)take(N * 8);
  SortArgs sa;
  sa.hist = (unsigned int*)take(K * 4);
  sa.offs = (unsigned int*)take((K + 1) * 4);
  sa.block_tot = (uint2*)take(n_blocks * 8);
  sa.packets = (int2*)take(max_packets * 8);
  sa.ctl = (unsigned int*)take(64);
  unsigned int* sorted = (unsigned int*)take(N * 4);
  unsigned int* live_list[2] = {(unsigned int*)take(N * 4), (unsigned int*)take(N * 4)};
  sa.n_keys = (int)K; sa.n_blocks = (int)n_blocks; sa.full_slices = s->bin_full_slices;
  ba.hist = sa.hist; ba.ctl = sa.ctl; ba.pool_run = s->bin_run;
  ba.scatter_p = 0; ba.dbg_base1 = s->bin_base1;
  if (!s->knobs.no_scatter) {
    const unsigned int nt = (unsigned int)local_tiles;
    unsigned int P = nt / 64 > 1 ? nt / 64 : 1;
    auto gcd = [](unsigned int x, unsigned int y) { while (y) { const unsigned int t = x % y; x = y; y = t; } return x; };
    while (gcd(P, nt) != 1) ++P;
    ba.scatter_p = (int)P;
  }
  BandArgs bd;
  bd.pool = s->pool; bd.blob = s->blob; bd.hdr = s->bin_hdr; bd.goff = s->bin_goff;
  bd.A0 = ba.A0; bd.A1 = ba.A1; bd.A5 = ba.A5; bd.slot = ba.slot; bd.offs = sa.offs; bd.packets = sa.packets; bd.ctl = sa.ctl; bd.sorted = sorted;
  bd.dense_min = 16; bd.full_slices = s->bin_full_slices;
  if (const char* e = std::getenv("PT_BAND_DENSE_MIN")) bd.dense_min = std::max(1, std::atoi(e)); // (experiments only)
  PT_HIP(hipMemsetAsync(sa.hist, 0, K * 4, st));
  PT_HIP(hipMemsetAsync(sa.ctl, 0, 64, st));
  const dim3 step_grid((unsigned int)((N + kBlock - 1) / kBlock)), step_block(kBlock);
  const int band_blocks = std::max(1, s->num_cus) * 6;
  auto step = [&](int gen) -> int {
    ba.gen = gen;
    ba.live_in = live_list[gen & 1]; ba.live_out = live_list[(gen + 1) & 1];
    if (s->has_image) {
      if (s->mats_simple) hipLaunchKernelGGL((bin_step_kernel<UV_WINNER, MATS_LAMB_LIGHT_SOLID>), step_grid, step_block, 0, st, ba);
      else hipLaunchKernelGGL((bin_step_kernel<UV_WINNER, MATS_ALL>), step_grid, step_block, 0, st, ba);
    } else {
      if (s->mats_simple) hipLaunchKernelGGL((bin_step_kernel<UV_NONE, MATS_LAMB_LIGHT_SOLID>), step_grid, step_block, 0, st, ba);
      else hipLaunchKernelGGL((bin_step_kernel<UV_NONE, MATS_ALL>), step_grid, step_block, 0, st, ba);
    }
    PT_HIP(hipGetLastError());
    hipLaunchKernelGGL(bin_count_kernel, dim3((unsigned int)n_blocks), dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(bin_prefix_kernel, dim3(1), dim3(64), 0, st, sa);
    hipLaunchKernelGGL(bin_offsets_kernel, dim3((unsigned int)n_blocks), dim3(1024), 0, st, sa);
    hipLaunchKernelGGL(bin_scatter_kernel, dim3((unsigned int)((N + 255) / 256)), dim3(256), 0, st, ba.A4, ba.A5, sa.offs, sorted, ba.live_out, sa.ctl);
    hipLaunchKernelGGL(band_kernel, dim3((unsigned int)band_blocks), dim3(64 * kBandWaves), 0, st, bd);
    PT_HIP(hipGetLastError());
    return PT_OK;
  };
  // Generations while they pay: the host reads the live count back every 8 generations (the first time after min(samples, 8)) and, once
  // fewer than `tail_frac` of the pixels are live, hands them to bin_finish_kernel.
  int gen = 0;
  unsigned int live = (unsigned int)N;
  const long long max_gen = (long long)p->samples * std::max(1, p->depth) + 2;
  double tail_frac = 0.25;
  if (const char* e = std::getenv("PT_BIN_TAIL")) tail_frac = std::atof(e); // (experiments only; 0: generations to the end)
  const unsigned int tail_at = (unsigned int)(tail_frac * (double)N);
  while (live != 0 && live >= tail_at && gen <= max_gen) {
    const int until = gen + 8;
    for (; gen < until; gen++) if (int rc = step(gen)) return rc;
    PT_HIP(hipMemcpyAsync(&live, sa.ctl + 3, sizeof live, hipMemcpyDeviceToHost, st));
    PT_HIP(hipStreamSynchronize(st));
  }
  if (live != 0 && gen <= max_gen) { // the tail: persistent waves finish what is left (ctl[3] live pixels in the list the last step wrote)
    ba.gen = gen; ba.live_in = live_list[gen & 1]; ba.live_out = live_list[(gen + 1) & 1];
    PT_HIP(hipMemsetAsync(sa.ctl + 5, 0, 4, st));
    int per_cu = 0;
    const void* fk = s->has_image ? (s->mats_simple ? (const void*)bin_finish_kernel<UV_WINNER, MATS_LAMB_LIGHT_SOLID> : (const void*)bin_finish_kernel<UV_WINNER, MATS_ALL>)
                                  : (s->mats_simple ? (const void*)bin_finish_kernel<UV_NONE, MATS_LAMB_LIGHT_SOLID> : (const void*)bin_finish_kernel<UV_NONE, MATS_ALL>);
    auto cached = s->occupancy.find(fk);
    if (cached != s->occupancy.end()) per_cu = cached->second;
    else { PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, fk, kBlock, 0)); s->occupancy[fk] = per_cu; }
    const unsigned int blocks = (unsigned int)std::min<long long>((long long)std::max(1, per_cu) * std::max(1, s->num_cus), ((long long)live + kBlock - 1) / kBlock * 64);
    const dim3 fgrid(std::max(1u, blocks)), fblock(kBlock);
    if (s->has_image) {
      if (s->mats_simple) hipLaunchKernelGGL((bin_finish_kernel<UV_WINNER, MATS_LAMB_LIGHT_SOLID>), fgrid, fblock, 0, st, ba);
      else hipLaunchKernelGGL((bin_finish_kernel<UV_WINNER, MATS_ALL>), fgrid, fblock, 0, st, ba);
    } else {
      if (s->mats_simple) hipLaunchKernelGGL((bin_finish_kernel<UV_NONE, MATS_LAMB_LIGHT_SOLID>), fgrid, fblock, 0, st, ba);
      else hipLaunchKernelGGL((bin_finish_kernel<UV_NONE, MATS_ALL>), fgrid, fblock, 0, st, ba);
    }
    PT_HIP(hipGetLastError());
    live = 0;
  }
  s->last_generations = gen;
  s->last_launch[0] = (int)step_grid.x; s->last_launch[1] = 64; s->last_launch[2] = 0; s->last_launch[3] = 0;
  if (live != 0) return fail(PT_ERR_HIP, "pt_render: the binned renderer did not finish within samples x depth generations");
  return PT_OK;
}

static int reserve_binned(const PtScene* s, size_t N, size_t K, size_t& bytes) {
  const size_t n_blocks = (K + 1023) / 1024, max_packets = (N / 64 + 1) * (size_t)s->bin_full_slices + std::min(K, N) + 1;
  auto r = [](size_t n) { return (n + 255) & ~(size_t)255; };
  bytes = 6 * r(N * 16) + r(N * 8) + r(K * 4) + r((K + 1) * 4) + r(n_blocks * 8) + r(max_packets * 8) + r(64) + 3 * r(N * 4);
  if (s->ws_bin && s->ws_bin_pixels >= N) return PT_OK;
  if (s->ws_bin) (void)hipFree(s->ws_bin);
  s->ws_bin = nullptr; s->ws_bin_pixels = 0;
  PT_HIP(hipMalloc(&s->ws_bin, bytes));
  s->ws_bin_pixels = N;
  return PT_OK;
}

static int launch_render(const PtScene* s, const PtCamera* cam, const PtRenderParams* p, float* fb, hipStream_t st) {
  int cur = -1;
  PT_HIP(hipGetDevice(&cur));
  if (cur != s->device)
    return fail(PT_ERR_INVALID_ARG, "pt_render: the scene lives on device " + std::to_string(s->device) + " but the current device is " +
                                        std::to_string(cur) + " (hipSetDevice to the scene's device first)");
  std::lock_guard<std::mutex> lock(s->sched);
  KArgs a;
  std::memcpy(&a.cam, cam, sizeof(Cam));
  a.blob = s->blob; a.mats = s->mats; a.pool = s->pool; a.atlas = s->atlas; a.fb = fb;
  a.n_runs = s->n_runs; a.blob_f4 = s->blob_f4; a.mats_f4 = s->mats_f4; a.n_hittables = s->n_hittables;
  a.width = p->width; a.height = p->height; a.samples = p->samples; a.depth = p->depth;
  a.inv_w = 1.0f / (float)p->width; a.inv_h = 1.0f / (float)p->height; // host IEEE division: correctly rounded
  a.pinhole = cam_is_pinhole(a.cam) ? 1 : 0;
  a.shard_index = p->shard_index; a.shard_count = p->shard_count;
  a.n_tiles = n_tiles_of(p, &a.tiles_x);
  const int local_tiles = (a.n_tiles - p->shard_index + p->shard_count - 1) / p->shard_count; // tiles this shard owns
  // pixels no lane owns (edge tiles, padded last tile, depth 0) read as 0
  PT_HIP(hipMemsetAsync(fb, 0, (size_t)pt_framebuffer_floats(p) * sizeof(float), st));
  if (local_tiles <= 0) return PT_OK;
  a.fast_ok = (s->fast_ok && !(p->flags & PT_FLAG_NO_FASTDIV)) ? 1 : 0;
  if (p->flags & PT_FLAG_SINGLE_STREAM) { // the reference's single-task executor: one sequential chain, one lane
    if (p->shard_count != 1) return fail(PT_ERR_INVALID_ARG, "PT_FLAG_SINGLE_STREAM needs shard_count == 1");
    if ((long long)p->width * p->height * p->samples > (1ll << 22))
      return fail(PT_ERR_TOO_LARGE, "PT_FLAG_SINGLE_STREAM is sequential by definition: width * height * samples <= 2^22");
    if (p->depth <= 0) return PT_OK;
    hipLaunchKernelGGL(render_single_stream_kernel, dim3(1), dim3(64), 0, st, a);
    PT_HIP(hipGetLastError());
    return PT_OK;
  }
  a.coop_prefix = (s->coop_ok && a.fast_ok && !(p->flags & PT_FLAG_NO_COOP)) ? s->coop_prefix : -1;
  if (s->binned && !(p->flags & (PT_FLAG_FORCE_STREAM | PT_FLAG_FAST_RNG))) return launch_binned(s, a, p, local_tiles, st);
  const size_t blob_bytes = (size_t)s->blob_f4 * 16;
  // a scene with a triangle pool is queried through per-lane loads from the global blob: the scalar-cache resident kernels,
  // whatever its size (PT_FLAG_FORCE_STREAM: the streaming kernel, which scans every triangle, as the A/B)
  const bool tri_pool = s->tri_pooled > 0 && !(p->flags & PT_FLAG_FORCE_STREAM);
  const bool resident = (blob_bytes <= kMaxLdsBlob || (p->flags & PT_FLAG_NO_LDS) || tri_pool) && !(p->flags & PT_FLAG_FORCE_STREAM);
  const bool lds = resident && !(p->flags & PT_FLAG_NO_LDS) && !tri_pool;
  a.n_local_pixels = local_tiles * PT_TILE_PIXELS;
  a.fast_chunks = 0; a.samples_total = p->samples; a.fast_stride = 0;
  long long launch_units = local_tiles; // waves worth of work in the queue (tiles; fast mode: tiles x chunks)
  // default: whole tiles for the resident kernels (coherent primary rays), single pixels for the lock-step
  // streaming kernel (a workgroup waits for its slowest lane); either can be forced
  // (the triangle-pool kernels' iterations are long and per-lane: single pixels, like the streaming kernel)
  a.heavy_pixels = 0; a.heavy_lanes = 64;
  a.tile_group = 64;
  if (const char* e = std::getenv("PT_TILE_GROUP")) { const int g = std::atoi(e); if (g == 16 || g == 32) a.tile_group = g; } // (experiment)
  a.tile_granular = (p->flags & PT_FLAG_TILE_GRANULAR) ? 1 : (p->flags & PT_FLAG_PIXEL_GRANULAR) ? 0 : ((resident && !tri_pool) ? 1 : 0);
  a.scatter_p = 0; a.scatter_log = 0;
  auto set_scatter = [&]() { // (after a.n_local_pixels is final: fast mode multiplies it by the chunks per pixel)
    if (!(tri_pool && !a.tile_granular && !s->knobs.no_scatter)) return;
    const unsigned int nt = (unsigned int)a.n_local_pixels >> 6;
    a.scatter_log = s->knobs.scatter_log;
    const unsigned int per_wave = 64u >> a.scatter_log; // tiles a wave's 64 pixels come from
    unsigned int P = nt / per_wave > 1 ? nt / per_wave : 1;
    auto gcd = [](unsigned int x, unsigned int y) { while (y) { const unsigned int t = x % y; x = y; y = t; } return x; };
    while (gcd(P, nt) != 1) ++P;
    a.scatter_p = (int)P;
  };
  set_scatter();
  a.cost = nullptr;
  a.cost_max = 0;
  a.tri_cache = nullptr;
  for (float& f : a.foot) f = 0.0f;
  a.resume_rng = nullptr; a.resume_spp = 0;
  a.prio_onset = 0; a.prio_t1 = a.prio_t2 = a.prio_t3 = 0;
  a.order = nullptr;
  const bool mlds = lds && blob_bytes + (size_t)s->mats_f4 * 16 <= kMaxLdsWithMaterials;
  const size_t shmem = lds ? blob_bytes + (mlds ? (size_t)s->mats_f4 * 16 : 0) : 0;
  a.n_split = nullptr;
  int n_waves_resident = 1;
  // Cooperative kernels (a ray's list split over idle lanes; the heaviest tiles rendered G lanes per pixel) cost ~15 % of
  // the ordinary-mode throughput, and the part of an iteration that cannot be split (camera, shading, ray context:
  // ~4 500 cycles of a lone wave's 13 200 on the Cornell-style scene) bounds what they can win: there G = 8 shortens a
  // pixel's chain by 1.4x for 4.6x the lane time, so they pay only where the scan of the list dominates an iteration
  // (496-hittable scene, shard 0/8: 522 -> 229 ms).
  // Sphere runs with a culling grid: the ordinary resident kernels walk it (a ray then tests tens of spheres instead of
  // hundreds: 496-hittable scene 1 660 -> 3 240 Msamples/s at 1080p), the cooperative kernels scan the lists with lane
  // groups.  With the grid as first tuned (margin 1.5 r) a lone wave's walk was no shorter than a group's list scan and small
  // frames / shards kept the cooperative kernels; since the retuning (margin 0.5 r, walk after the big spheres) the walk wins
  // everywhere (tools/grid_min_tiles.py: 400x225x256 spp 99 ms against 108; shard 0/8 of the 1080p frame at 256 spp 80
  // against 102; shard 0/8 of 4K at 128 spp 61 against 102).  PT_GRID_MIN_TILES restores a threshold.
  const bool use_grid = s->grid_spheres > 0 && resident /* the streaming kernel scans the full lists */ && local_tiles >= s->knobs.grid_min_tiles && !(p->flags & PT_FLAG_FORCE_COOP);
  const bool coop = lds && a.coop_prefix >= 0 && !use_grid && (s->traversal_cost >= kCoopMinTraversal || (p->flags & PT_FLAG_FORCE_COOP));
  // Which sphere-grid walk (pt_device.hpp: sphere_scan, GRID = 1 / 2): the pair queue where walks diverge or the launch is bound by its
  // chains — whole tiles per wave or at least 16 lanes (the rule of `launch` below, evaluated for the kernels' four workgroups per CU) — and
  // not on frames that are both dense (>= 4 M pixels: a tile's rays share their cells) and throughput-bound (>= 6 pixels per resident lane).
  // The rule is evaluated for the resident lanes the GRID = 2 kernels really have (their occupancy is queried like `launch` does: 31 KB LDS
  // image + 4 KB of per-wave queues and slots usually allow four workgroups per CU) — round 4 assumed 16 waves per CU here and the cap of
  // `launch` from the measured occupancy, and the two could disagree.  The queued walk's STATIC 4 KB of LDS (sphere_queue,
  // sphere_slots) ride on top of the dynamic blob image: where the sum exceeds what a workgroup may have, the walk stays in place.
  constexpr size_t kQueuedWalkStaticLds = (size_t)PT_MAX_WAVES_PER_BLOCK * (PT_SQ_CAP * 4 + 64 * 8);
  bool queued_walk = false;
  auto decide_walk = [&](int per_cu_queued, int waves_per_block) {
    if (!use_grid) return;
    const bool fits = shmem + kQueuedWalkStaticLds <= s->lds_per_block && per_cu_queued >= 1;
    const double lanes = (double)std::max(1, s->num_cus) * std::max(1, per_cu_queued) * waves_per_block * 64.0;
    const double rho = (double)a.n_local_pixels / lanes;
    const bool few_lanes = rho < 0.875 && !(p->flags & (PT_FLAG_TILE_GRANULAR | PT_FLAG_PIXEL_GRANULAR | PT_FLAG_FAST_RNG)) && s->knobs.lanes_cap >= 0; // 16 rho rounds below 16
    const bool dense_and_busy = (double)p->width * p->height >= 4.0e6 && rho >= 6.0;
    queued_walk = fits && (s->knobs.grid_walk == 2 || (s->knobs.grid_walk == 0 && !few_lanes && !dense_and_busy && !(s->knobs.lanes_cap > 0 && s->knobs.lanes_cap < 16)));
  };
  // (small frames through the grid kernels: see `launch`; PT_FLAG_TILE_GRANULAR / PT_FLAG_PIXEL_GRANULAR and the fast mode keep what they ask for)
  // (only the kernels that walk a sphere grid: the headline family's iteration — slab pass, wave-uniform — does not get shorter with fewer
  // lanes: its shard 0/8 took 127 ms that way against 40)
  const bool share_small = use_grid && s->knobs.lanes_cap >= 0 && !(p->flags & (PT_FLAG_TILE_GRANULAR | PT_FLAG_PIXEL_GRANULAR | PT_FLAG_FAST_RNG));
  // (set by launch_uv for the headline family — no sphere grid, no cooperative phase: see `launch`)
  bool chain_bound_family = false;
  bool launched_queued_walk = false; // the kernel variant actually launched walks sphere grids through the pair queue (pt_debug_last_launch)
  // Persistent grid: no more workgroups than the chip holds at once; lanes pull pixels from the queue.
  auto launch = [&](auto kernel, int block_threads = kBlock) -> int {
    const int waves_per_block = block_threads / 64;
    int per_cu = 0;
    auto cached = s->occupancy.find((const void*)kernel);
    if (cached != s->occupancy.end()) per_cu = cached->second;
    else {
      PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kernel, block_threads, shmem));
      s->occupancy[(const void*)kernel] = per_cu;
    }
    // (a kernel variant that does not fit a CU at all — its static + dynamic LDS beyond the device's limit — must not be launched and hoped for)
    if (per_cu < 1) return fail(PT_ERR_TOO_LARGE, "pt_render: this scene's kernel variant does not fit a compute unit (LDS " + std::to_string(shmem) + " B dynamic + static)");
    // A headline-family launch with fewer than ~1.15 tiles per wave slot (shard 0 of 4 ... 6 of the 1080p frame) is bound by its heaviest
    // tiles' sequential chains, and a chain's iteration takes as long as the waves that share its SIMD make it: four waves per SIMD that
    // take two tiles each, heaviest first, finish before eight that take one.  Cornell-style 1080p x 1024 spp, kernel ms of shard 0 of
    // N = 3 / 4 / 6 at 8 and at 4 workgroups per CU: 63.3 / 57.8 / 43.4 and 60.3 / 51.1 / 42.8 (profiles/r04_blocks_sweep.txt); whole frames
    // and halves (>= 2 tiles per slot) keep the full occupancy, and a launch with half a tile per slot takes four per CU by itself.
    // (Grid and triangle-pool kernels: their own occupancy is the best at every shard count — same file.)
    // (Round 5, with the chain priorities of render_kernel: shard 0 of 3 — 1.3 tiles per slot — 61.2 ms at four workgroups per CU, 54.4 / 57.1 at
    // six / eight; shard 0 of 4 — 1.0 — 45.2 / 47.3 / 48.7: the bound moved from 1.6 to 1.15; profiles/r05_ab_chain_priority.txt.)
    if (chain_bound_family && !s->knobs.blocks_per_cu && a.n_split == 0 && a.scatter_p == 0 &&
        (double)launch_units < 1.15 * (double)per_cu * waves_per_block * std::max(1, s->num_cus))
      per_cu = std::min(per_cu, 4);
    if (s->knobs.blocks_per_cu) per_cu = std::min(per_cu, s->knobs.blocks_per_cu); // tuning knob
    // longest remaining chain first (render_kernel): the headline family's frame launches in parity mode, from the middle of the queue on
    a.prio_onset = 0; a.prio_t1 = a.prio_t2 = a.prio_t3 = 0;
    if (chain_bound_family && !a.cost && !a.fast_chunks && a.tile_granular && !s->knobs.no_chain_prio && p->samples >= 16) {
      a.prio_onset = a.n_local_pixels / 2;
      a.prio_t3 = p->samples / 2; a.prio_t2 = p->samples / 4; a.prio_t1 = p->samples / 8;
    }
    const int resident_blocks = std::max(1, per_cu) * std::max(1, s->num_cus);
    // Queue counters come from a ring of kQueueRing slots.  A slot is reused only after the launch that last used it has
    // finished (more than kQueueRing launches in flight on one scene would otherwise share a dequeue counter and lose or
    // duplicate pixels): the NEW launch's stream waits for that launch's event on the device — the host does not block, so
    // pt_render stays asynchronous however many renders are queued (round 3 synchronised the host here, with the scene's
    // scheduling mutex held).
    const unsigned int slot = s->next_queue % kQueueRing;
    if (s->ring_done[slot]) PT_HIP(hipStreamWaitEvent(st, s->ring_done[slot], 0));
    else PT_HIP(hipEventCreateWithFlags(&s->ring_done[slot], hipEventDisableTiming));
    s->next_queue++; // (only once nothing above can fail any more)
    a.queue = s->queues + 2 * slot; // [0] ordinary queue, [1] wide-phase queue
    PT_HIP(hipMemsetAsync(a.queue, 0, 2 * sizeof(unsigned int), st));
    // one wave per tile is enough, except in the wide phase, where a split tile keeps G waves busy (how many tiles are
    // split is decided on the device, so such a launch simply fills the chip; surplus waves find the queues empty and exit)
    long long wanted = a.n_split ? (long long)resident_blocks : (launch_units + waves_per_block - 1) / waves_per_block;
    a.lanes_cap = 64;
    // Frames (or shards) that do not fill the chip several times over are bound by their pixels' sequential chains, not by throughput —
    // BASELINE config 1 is 1 406 tiles for 4 096 resident waves, a shard of an 8-GPU job about one tile per wave — and a wave that steps
    // 64 pixels together pays, every iteration, for the LONGEST walk, the largest candidate count and every material among them.  Such a
    // launch fills the chip and gives each wave only `lanes_cap` lanes' worth of pixels at a time (one pixel per lane, refilled from the
    // queue: consecutive positions = neighbouring pixels of one tile row, so the walks stay coherent): fewer lanes per wave = a shorter
    // iteration for the pixels that set the frame time.  Measured on one box (profiles/r04_ab_lanes_cap.txt; rho = pixels per resident
    // lane): config 1 (rho 0.34) 24.5 ms at 64 lanes, 22.0 at 24, 19.2 at 8, 17.9 at 4; shard 0/8 of the 1080p frame (rho 1.0) 70.9 / 61.6 /
    // 60.1 / 76.4 ms at 64 / 32 / 16 / 8; shard 0/8 of the 4K frame (rho 4.0) 201.7 / 209.8 / 218.1 at 64 / 48 / 32: lanes = 16 rho fits all
    // three (rounded to the nearest multiple of 4).  Which lane renders a pixel changes nothing in the image (a pixel's seed is its id).
    if (share_small && a.scatter_p == 0) {
      const long long lanes = (long long)resident_blocks * waves_per_block * 64;
      long long cap = (16 * (long long)a.n_local_pixels + lanes / 2) / lanes; // 16 rho
      cap = std::min<long long>(64, std::max<long long>(4, (cap + 2) / 4 * 4)); // (the nearest multiple of 4: half a tile row)
      if (cap > 24) cap = 64; // (rho 2: 93.9 ms at 32 lanes against 90.9 at 64 — from there on the frame is throughput)
      if (s->knobs.lanes_cap > 0) cap = std::min(64, s->knobs.lanes_cap); // PtTuning.lanes_cap: forced
      if (cap < 64) {
        a.lanes_cap = (int)cap;
        a.tile_granular = 0;
        wanted = std::min<long long>(resident_blocks, ((long long)a.n_local_pixels + cap * waves_per_block - 1) / (cap * waves_per_block));
        wanted = std::max<long long>(wanted, 1);
      }
    }
    // Between the launches that the rule above serves (fewer than 1.5 pixels per resident lane) and the ones that are throughput (6 and
    // more) lie the shards of a multi-GPU job on a big frame — one of 8 GPUs on the 4K frame: 4 pixels per lane — whose time is their
    // heaviest tiles' chains: whole tiles everywhere, except that the head of the cost-sorted queue (one tile per SIMD of the chip) is handed
    // out 16 pixels at a time (lane_acquire: heavy_pixels).  Kernel ms of shard 0 of N, whole tiles / with the narrow head — 4K x 512 spp
    // N = 8 / 6 / 5 / 4: 196.8 / 203.9 / 220.4 / 209.1 -> 162.7 / 169.1 / 217.2 / 208.8; 1080p x 512 spp N = 2 / 3 / 4: 189.3 / 189.2 / 176.2 ->
    // 165.4 / 159.2 / 155.3; whole frames lose (1080p: 382 -> 435 ms), hence the upper bound (profiles/r04_heavy_tiles.txt).
    // Needs the cost-sorted order (the probe pass ran) — and changes nothing in the image: a pixel's seed is its id.
    if (share_small && a.lanes_cap == 64 && a.tile_granular && a.order && a.scatter_p == 0 && !a.cost && s->knobs.heavy_tiles >= 0) {
      const double lanes = (double)resident_blocks * waves_per_block * 64.0, rho = (double)a.n_local_pixels / lanes;
      long long tiles = s->knobs.heavy_tiles > 0 ? s->knobs.heavy_tiles : (rho >= 1.5 && rho < 6.0 ? 4LL * std::max(1, s->num_cus) : 0);
      tiles = std::min<long long>(tiles, a.n_local_pixels / 64 / 2);
      if (tiles > 0) { a.heavy_pixels = (int)(tiles * 64); a.heavy_lanes = 16; }
    }
    if (a.scatter_p > 0) { // triangle-pool kernels: fill the chip and share the pixels out evenly (lane_acquire)
      wanted = std::min<long long>(resident_blocks, ((long long)a.n_local_pixels + block_threads - 1) / block_threads * 64); // (at least one pixel per wave)
      wanted = std::max<long long>(wanted, 1);
      const long long waves = std::min<long long>(wanted, resident_blocks) * waves_per_block;
      a.lanes_cap = (int)std::min<long long>(64, std::max<long long>(1, ((long long)a.n_local_pixels + waves - 1) / waves));
    }
    n_waves_resident = (int)std::min<long long>(wanted, resident_blocks) * waves_per_block;
    if (!a.cost) { s->last_launch[0] = (int)std::min<long long>(wanted, resident_blocks); s->last_launch[1] = a.lanes_cap; s->last_launch[2] = a.heavy_pixels; s->last_launch[3] = launched_queued_walk ? 1 : 0; }
    dim3 grid((unsigned int)std::min<long long>(wanted, resident_blocks)), block((unsigned int)block_threads);
    // The camera rays' candidate cache (pt_device.hpp: TriPrimCtx): triangle-pool kernels, a pinhole camera (the rays of a pixel share
    // their origin), pixel coordinates that fit 16 bits.  One 1 KB line per lane of the launch; every line starts out belonging to no pixel.
    a.tri_cache = nullptr;
    if (tri_pool && a.cam.lens_radius == 0.0f && !s->knobs.no_tri_cache && p->width <= 65535 && p->height <= 65535) {
      const size_t lanes = (size_t)grid.x * block.x;
      if (s->ws_tricache_lanes < lanes) {
        if (s->ws_tricache) (void)hipFree(s->ws_tricache);
        s->ws_tricache = nullptr; s->ws_tricache_lanes = 0;
        PT_HIP(hipMalloc((void**)&s->ws_tricache, lanes * PT_TRI_CACHE_WORDS * 4));
        s->ws_tricache_lanes = lanes;
      }
      PT_HIP(hipMemsetAsync(s->ws_tricache, 0xff, lanes * PT_TRI_CACHE_WORDS * 4, st));
      a.tri_cache = s->ws_tricache;
      // the footprint of a pixel in direction space (camera.hpp:93-100 with lens_radius 0: d = llc + s hor + t ver - origin, s in [x / W, (x + 1) / W])
      const Cam& cm = a.cam;
      float base[3], hw[3], vh[3];
      double dmax = 0, lh = 0, lv = 0;
      for (int k = 0; k < 3; k++) {
        base[k] = cm.llc[k] - cm.origin[k]; hw[k] = cm.horizontal[k] * a.inv_w; vh[k] = cm.vertical[k] * a.inv_h;
        a.foot[k] = base[k]; a.foot[3 + k] = hw[k]; a.foot[6 + k] = vh[k];
        lh += (double)hw[k] * hw[k]; lv += (double)vh[k] * vh[k];
        dmax += std::fabs((double)cm.llc[k]) + std::fabs((double)cm.origin[k]) + std::fabs((double)cm.horizontal[k]) + std::fabs((double)cm.vertical[k]);
      }
      lh = std::sqrt(lh); lv = std::sqrt(lv);
      // half a pixel each way (the jitter may round to the pixel's far edge), + the binary32 rounding of the camera's own arithmetic and of the centre ray
      a.foot[9] = (float)((0.5 * (lh + lv)) * 1.001 + 1e-5 * dmax);
    }
    hipLaunchKernelGGL(kernel, grid, block, shmem, st, a);
    PT_HIP(hipGetLastError());
    PT_HIP(hipEventRecord(s->ring_done[slot], st));
    return PT_OK;
  };
  auto launch_uv = [&](auto uv) -> int {
    constexpr int UV = decltype(uv)::value;
    if (s->has_badouel) { // parity-completeness path: scalar-cache or streaming kernel with the Badouel loop compiled in
      if (a.fast_chunks) return fail(PT_ERR_INVALID_ARG, "PT_FLAG_FAST_RNG is not offered for scenes with Badouel-strategy triangles");
      if (!resident) return launch(render_kernel_stream<UV, false, true>);
      return launch(render_kernel<UV, false, false, false, false, false, true>);
    }
    if (a.fast_chunks) { // opt-in decorrelated mode: its own instantiations (no cooperative kernels: a chunk is short)
      if (tri_pool) return launch(render_kernel<UV, false, false, false, false, true, false, true, true>);
      if (!resident) return launch(render_kernel_stream<UV, true>);
      if (!lds) return launch(render_kernel<UV, false, false, false, false, true>);
      return mlds ? launch(render_kernel<UV, true, true, false, false, true>) : launch(render_kernel<UV, true, false, false, false, true>);
    }
    if (tri_pool) return launch(render_kernel<UV, false, false, false, false, false, false, true, true>);
    if (!resident) return launch(render_kernel_stream<UV>);
    if constexpr (UV == UV_NONE) { // no image texture and no sphere grid (the headline scene): kernels without the grid walk
      if (s->grid_spheres == 0 && !coop) {
        chain_bound_family = true;
        if (s->mats_simple) { // lambertian + lightsource over solid textures: kernels without the other materials' code
          if (s->rectbox_only) { // ... and every hittable a rect or a box (the headline scene): resolve_hit without the other kinds
            constexpr int MSR = MATS_LAMB_LIGHT_SOLID | MATS_RECTBOX_ONLY;
            if (!lds) return launch(render_kernel<UV, false, false, false, false, false, false, false, false, MSR>);
            if (mlds && shmem <= kMaxLdsColdScene && !s->knobs.no_cold_lds) return launch(render_kernel<UV, true, true, false, true, false, false, false, false, MSR>);
            return mlds ? launch(render_kernel<UV, true, true, false, false, false, false, false, false, MSR>)
                        : launch(render_kernel<UV, true, false, false, false, false, false, false, false, MSR>);
          }
          constexpr int MS = MATS_LAMB_LIGHT_SOLID;
          if (!lds) return launch(render_kernel<UV, false, false, false, false, false, false, false, false, MS>);
          if (mlds && shmem <= kMaxLdsColdScene && !s->knobs.no_cold_lds) return launch(render_kernel<UV, true, true, false, true, false, false, false, false, MS>);
          return mlds ? launch(render_kernel<UV, true, true, false, false, false, false, false, false, MS>)
                      : launch(render_kernel<UV, true, false, false, false, false, false, false, false, MS>);
        }
        if (!lds) return launch(render_kernel<UV, false, false, false, false, false, false, false>);
        if (mlds && shmem <= kMaxLdsColdScene && !s->knobs.no_cold_lds) return launch(render_kernel<UV, true, true, false, true, false, false, false>);
        return mlds ? launch(render_kernel<UV, true, true, false, false, false, false, false>)
                    : launch(render_kernel<UV, true, false, false, false, false, false, false>);
      }
    }
    if (!lds) return launch(render_kernel<UV, false, false, false>);
    if (coop) return mlds ? launch(render_kernel<UV, true, true, true>) : launch(render_kernel<UV, true, false, true>);
    if constexpr (UV == UV_NONE) { // small scene: cold lane state in LDS (7 workgroups x (scene + 8 KB) per CU)
      if (mlds && shmem <= kMaxLdsColdScene && !s->knobs.no_cold_lds) return launch(render_kernel<UV, true, true, false, true>);
    }
    if (use_grid) { // which walk: decided with the queued-walk kernel's own occupancy
      int per_cu2 = 0;
      const void* k2 = mlds ? (const void*)render_kernel<UV, true, true, false, false, false, false, 2> : (const void*)render_kernel<UV, true, false, false, false, false, false, 2>;
      auto cached = s->occupancy.find(k2);
      if (cached != s->occupancy.end()) per_cu2 = cached->second;
      else if (shmem + kQueuedWalkStaticLds <= s->lds_per_block) {
        PT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu2, k2, kBlock, shmem));
        s->occupancy[k2] = per_cu2;
      }
      decide_walk(s->knobs.blocks_per_cu ? std::min(per_cu2, s->knobs.blocks_per_cu) : per_cu2, kWavesPerBlock);
    }
    if (use_grid && queued_walk) { // the sphere-grid walk through the pair queue (pt_device.hpp: sphere_scan says where it pays)
      launched_queued_walk = true;
      return mlds ? launch(render_kernel<UV, true, true, false, false, false, false, 2>) : launch(render_kernel<UV, true, false, false, false, false, false, 2>);
    }
    return mlds ? launch(render_kernel<UV, true, true, false>) : launch(render_kernel<UV, true, false, false>);
  };
  auto launch_variant = [&]() -> int {
    if (s->track_uv) return launch_uv(std::integral_constant<int, UV_TRACKED>{});
    if (s->has_image) return launch_uv(std::integral_constant<int, UV_WINNER>{});
    return launch_uv(std::integral_constant<int, UV_NONE>{});
  };
  // Heaviest-first tile order from a probe pass (see lpt_order_kernel); pointless for short renders.
  int probe_spp = std::min(4, p->samples / 16);
  // Tiles are ordered by their ray count — or, where a wave holds a tile until its last pixel is done and pixels differ
  // widely (the sphere-field scenes the grid kernels run: a few glass / mirror pixels per tile), by 64 x their heaviest
  // pixel's, from a deeper probe: 496-hittable scene +3-6 % (same box), Cornell-style -0.8 % (kept on the sum).  (Round 5, with the
  // probe's samples kept: still samples / 64 up to 16 — 1080p x 1024 spp 448 / 409 / 395 / 375 ms at depths 2 / 4 / 8 / 16 and 379 / 387 /
  // 406 at 32 / 64 / 128: an unordered launch runs at half the ordered one's pace; profiles/r05_ab_probe_resume.txt.)
  const bool cost_by_max = s->knobs.lpt_max >= 0 ? s->knobs.lpt_max != 0 : use_grid;
  if (cost_by_max) probe_spp = std::min(std::max(probe_spp, p->samples / 64), s->knobs.probe_spp_max);
  // (the triangle-pool kernels deal every wave a stratified sample of the frame's tiles — lane_acquire, scatter_p — which balances the
  // waves without knowing the costs: the probe only costs there, 1080p x 32 spp 3.12 -> 2.95 s without it; PT_LPT_SCATTER=1 keeps it)
  const bool probe_pays = !(a.scatter_p > 0) || s->knobs.lpt_with_scatter;
  if (probe_spp >= 1 && local_tiles >= 64 && !(p->flags & PT_FLAG_NO_LPT) && probe_pays) {
    if (int rc = reserve_tiles(s, local_tiles)) return rc; // first render at a new size only (or never: pt_scene_reserve)
    PT_HIP(hipMemsetAsync(s->ws_cost, 0, (size_t)local_tiles * sizeof(unsigned int), st));
    KArgs main_args = a;
    a.cost = s->ws_cost;
    a.cost_max = (cost_by_max && !coop) ? 1 : 0;
    a.samples = probe_spp;
    // the probe's samples are kept (KArgs.resume_rng) — not in the opt-in fast mode, whose chunks are streams of their own
    const bool resume = !(p->flags & PT_FLAG_FAST_RNG) && !s->knobs.no_resume;
    a.resume_rng = resume ? s->ws_rng : nullptr;
    int rc = launch_variant();
    if (rc) return rc;
    // rough per-iteration instruction counts: traversal (splittable) vs shading + camera + cooperative overhead (not)
    const int forced_logG = s->knobs.wide_logG; // 0: the model picks the group size of the wide phase
    const unsigned int* ranked = s->ws_cost;
    if (a.cost_max && p->shard_count == 1 && !coop) { // tile_dilate_kernel (a shard's neighbours are other ranks' tiles: its own estimates)
      hipLaunchKernelGGL(tile_dilate_kernel, dim3((unsigned int)((local_tiles + 255) / 256)), dim3(256), 0, st, s->ws_cost, a.tiles_x, (int)(a.n_tiles / a.tiles_x), s->ws_rank);
      PT_HIP(hipGetLastError());
      ranked = s->ws_rank;
    }
    // rough per-iteration instruction counts: traversal (splittable) vs shading + camera (not)
    hipLaunchKernelGGL(lpt_order_kernel, dim3(1), dim3(1024), 0, st, ranked, local_tiles, s->ws_order, n_waves_resident,
                       std::max(1.0f, s->traversal_cost), s->knobs.model_fixed, s->knobs.model_chain, forced_logG, coop ? s->ws_nsplit : nullptr);
    PT_HIP(hipGetLastError());
    if (s->knobs.has_split_tiles) { // tuning knob: fixed number of split tiles (< 0: all)
      if (coop) {
        const int k = s->knobs.split_tiles;
        s->nsplit_override = k < 0 ? local_tiles : std::min(k, local_tiles);
        PT_HIP(hipMemcpyAsync(s->ws_nsplit, &s->nsplit_override, sizeof(int), hipMemcpyHostToDevice, st));
      }
    }
    // (A second, ORDERED probe stage of samples / 8 that keeps counting also removes the slow launches tile_dilate_kernel is there for — 1080p x
    // 1024 spp 375 / 407 -> 380.8 +- 1.5 — at the price of the fast ones and of 4 % on chain-bound shards: not kept.)
    a = main_args;
    if (resume) { a.resume_rng = s->ws_rng; a.resume_spp = probe_spp; }
    a.order = s->ws_order;
    a.n_split = (coop && !(p->flags & PT_FLAG_NO_SPLIT)) ? s->ws_nsplit : nullptr;
  }
  if (p->flags & PT_FLAG_FAST_RNG) {
    // OPT-IN decorrelated mode (include/pt_render.h): (tile, chunk) work units, per-chunk sums into a workspace, then
    // one ordered reduction.  No wide phase (a chunk is short by construction); the tile order of the probe still applies.
    const int chunks = (p->samples + PT_FAST_CHUNK_SPP - 1) / PT_FAST_CHUNK_SPP;
    if (chunks > 127 || local_tiles >= (1 << 18))
      return fail(PT_ERR_TOO_LARGE, "PT_FLAG_FAST_RNG supports up to 8128 samples per pixel and 2^24 pixels per shard");
    const size_t plane = (size_t)pt_framebuffer_floats(p), need = plane * (size_t)chunks;
    if (int rc = reserve_partial(s, need)) return rc; // first render at a new size only (or never: pt_scene_reserve)
    PT_HIP(hipMemsetAsync(s->ws_partial, 0, need * sizeof(float), st)); // pixels no lane owns add 0
    a.fast_chunks = chunks;
    a.samples = std::min(p->samples, (int)PT_FAST_CHUNK_SPP);
    a.samples_total = p->samples;
    a.fast_stride = (long long)plane;
    a.fb = s->ws_partial;
    a.n_local_pixels = local_tiles * PT_TILE_PIXELS * chunks;
    set_scatter();
    a.n_split = nullptr;
    launch_units = (long long)local_tiles * chunks;
    s->last_had_wide_phase = false;
    int rc = launch_variant();
    if (rc) return rc;
    hipLaunchKernelGGL(fast_reduce_kernel, dim3((unsigned int)((plane + 255) / 256)), dim3(256), 0, st, s->ws_partial, fb,
                       (long long)plane, (long long)plane, chunks, (float)p->samples);
    PT_HIP(hipGetLastError());
    return PT_OK;
  }
  s->last_had_wide_phase = a.n_split != nullptr;
  int rc = launch_variant();
  if (rc) return rc;
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int pt_scene_reserve(const PtScene* scene, const PtRenderParams* p) {
  if (!scene) return fail(PT_ERR_INVALID_ARG, "pt_scene_reserve: NULL scene");
  int rc = check_params(p);
  if (rc) return rc;
  int cur = -1;
  PT_HIP(hipGetDevice(&cur));
  if (cur != scene->device) return fail(PT_ERR_INVALID_ARG, "pt_scene_reserve: the scene lives on another device");
  std::lock_guard<std::mutex> lock(scene->sched);
  int tiles_x;
  const int n_tiles = n_tiles_of(p, &tiles_x);
  const int local_tiles = (n_tiles - p->shard_index + p->shard_count - 1) / p->shard_count;
  if (local_tiles > 0 && (rc = reserve_tiles(scene, local_tiles))) return rc;
  if (p->flags & PT_FLAG_FAST_RNG) {
    const int chunks = (p->samples + PT_FAST_CHUNK_SPP - 1) / PT_FAST_CHUNK_SPP;
    if ((rc = reserve_partial(scene, (size_t)pt_framebuffer_floats(p) * (size_t)chunks))) return rc;
  }
  return PT_OK;
}

int pt_render(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p, float* fb_device, void* stream) {
  if (!scene || !cam || !fb_device) return fail(PT_ERR_INVALID_ARG, "pt_render: NULL argument");
  int rc = check_params(p);
  if (rc) return rc;
  return launch_render(scene, cam, p, fb_device, (hipStream_t)stream);
}

int pt_render_timed(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p, float* fb_device, void* stream,
                    float* kernel_ms) {
  if (!scene || !cam || !fb_device || !kernel_ms) return fail(PT_ERR_INVALID_ARG, "pt_render_timed: NULL argument");
  int rc = check_params(p);
  if (rc) return rc;
  hipStream_t st = (hipStream_t)stream;
  EventPair ev;
  PT_HIP(hipEventCreate(&ev.e0));
  PT_HIP(hipEventCreate(&ev.e1));
  PT_HIP(hipEventRecord(ev.e0, st));
  rc = launch_render(scene, cam, p, fb_device, st);
  if (rc != PT_OK) return rc;
  PT_HIP(hipEventRecord(ev.e1, st));
  PT_HIP(hipEventSynchronize(ev.e1));
  PT_HIP(hipEventElapsedTime(kernel_ms, ev.e0, ev.e1));
  return PT_OK;
}

int pt_render_host(const PtScene* scene, const PtCamera* cam, const PtRenderParams* p, float* fb_host) {
  if (!scene || !cam || !fb_host) return fail(PT_ERR_INVALID_ARG, "pt_render_host: NULL argument");
  int rc = check_params(p);
  if (rc) return rc;
  size_t bytes = (size_t)pt_framebuffer_floats(p) * sizeof(float);
  float* d = nullptr;
  PT_HIP(hipMalloc((void**)&d, bytes));
  rc = launch_render(scene, cam, p, d, nullptr);
  if (rc == PT_OK) {
    hipError_t e = hipMemcpy(fb_host, d, bytes, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(PT_ERR_HIP, std::string("hipMemcpy D2H: ") + hipGetErrorString(e));
  }
  (void)hipFree(d);
  return rc;
}

int pt_unshard_tiles(const float* gathered_device, const PtRenderParams* p, float* fb_device, void* stream) {
  if (!gathered_device || !fb_device) return fail(PT_ERR_INVALID_ARG, "pt_unshard_tiles: NULL argument");
  int rc = check_params(p);
  if (rc) return rc;
  int tiles_x;
  int n_tiles = n_tiles_of(p, &tiles_x);
  int per = (n_tiles + p->shard_count - 1) / p->shard_count;
  dim3 block(256), grid((p->width + 255) / 256, p->height);
  hipLaunchKernelGGL(unshard_kernel, grid, block, 0, (hipStream_t)stream, gathered_device, fb_device, p->width, p->height,
                     tiles_x, p->shard_count, per);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

int pt_tonemap_rgb8(const float* fb_device, int32_t width, int32_t height, uint8_t* rgb8_device, void* stream) {
  if (!fb_device || !rgb8_device || width <= 0 || height <= 0) return fail(PT_ERR_INVALID_ARG, "pt_tonemap_rgb8: bad argument");
  dim3 block(256), grid((width + 255) / 256, height);
  hipLaunchKernelGGL(tonemap_kernel, grid, block, 0, (hipStream_t)stream, fb_device, rgb8_device, width, height);
  PT_HIP(hipGetLastError());
  return PT_OK;
}

#ifdef PT_BIN_DEBUG
extern "C" int pt_debug_bin_fallbacks(unsigned int* out) { // -DPT_BIN_DEBUG: per generation (rays, waves) that scanned the pooled run in full
  PT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_bin_fb), 4096 * sizeof(unsigned int)));
  return PT_OK;
}
#endif
#ifdef PT_STAMPS_BLOCKS
extern "C" int pt_debug_blocks(unsigned long long* out, int n_blocks) { // -DPT_STAMPS_BLOCKS: (where, start, end) of the last frame launch's workgroups
  PT_HIP(hipMemcpyFromSymbol(out, HIP_SYMBOL(g_blocks), (size_t)std::min(n_blocks, 8192) * 3 * sizeof(unsigned long long)));
  return PT_OK;
}
#endif
#ifdef PT_STAMPS
int pt_debug_stamps(unsigned long long* out8, int reset) { // diagnostic build only; not part of include/pt_render.h
  if (out8) PT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), 8 * sizeof(unsigned long long)));
  if (reset) { unsigned long long z[8] = {0}; PT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof z)); }
  return PT_OK;
}
#ifdef PT_STAMPS_TRI
int pt_debug_tri(unsigned long long* out8, int reset) { // -DPT_STAMPS_TRI: counters of the triangle pool (pt_device.hpp: PT_TRI_COUNT)
  if (out8) PT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tri), 16 * sizeof(unsigned long long)));
  if (reset) { unsigned long long z[16] = {0}; PT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_tri), z, sizeof z)); }
  return PT_OK;
}
#endif
#ifdef PT_STAMPS_RUNS
int pt_debug_runs(unsigned long long* out16, int reset) { // -DPT_STAMPS_RUNS: cycles per run of the hittable list (pt_device.hpp: hit_world)
  if (out16) PT_HIP(hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_runs), 16 * sizeof(unsigned long long)));
  if (reset) { unsigned long long z[16] = {0}; PT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_runs), z, sizeof z)); }
  return PT_OK;
}
#endif
#ifdef PT_STAMPS_WALK
int pt_debug_walk(unsigned long long* out8, int reset) { // -DPT_STAMPS_WALK: counters of the sphere-grid walk (pt_device.hpp: walk_ctr)
  if (out8) PT_HIP(hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_walk), 8 * sizeof(unsigned long long)));
  if (reset) { unsigned long long z[8] = {0}; PT_HIP(hipMemcpyToSymbol(HIP_SYMBOL(g_walk), z, sizeof z)); }
  return PT_OK;
}
#endif
#endif

// ---- probes: host arrays in/out ------------------------------------------------------------------------

int pt_debug_bounce(const PtScene* scene, const PtBounceIn* in, PtBounceOut* out, int32_t n) {
  if (!scene || !in || !out || n < 0) return fail(PT_ERR_INVALID_ARG, "pt_debug_bounce: bad argument");
  if (n == 0) return PT_OK;
  DevBuf<PtBounceIn> din;
  DevBuf<PtBounceOut> dout;
  PT_HIP(din.alloc(n));
  PT_HIP(dout.alloc(n));
  PT_HIP(hipMemcpy(din.p, in, (size_t)n * sizeof(PtBounceIn), hipMemcpyHostToDevice));
  dim3 block(64), grid((n + 63) / 64);
  const bool qw = scene->knobs.grid_walk == 2; // PtTuning.grid_walk = 2: the probe walks sphere grids through the pair queue, like the kernels it stands for
  if (scene->track_uv && qw)
    hipLaunchKernelGGL((bounce_kernel<true, 2>), grid, block, 0, nullptr, scene->blob, scene->n_runs, scene->mats, scene->pool, scene->atlas, din.p, dout.p, n, scene->fast_ok ? 1 : 0);
  else if (scene->track_uv)
    hipLaunchKernelGGL((bounce_kernel<true, 1>), grid, block, 0, nullptr, scene->blob, scene->n_runs, scene->mats, scene->pool, scene->atlas, din.p, dout.p, n, scene->fast_ok ? 1 : 0);
  else if (qw)
    hipLaunchKernelGGL((bounce_kernel<false, 2>), grid, block, 0, nullptr, scene->blob, scene->n_runs, scene->mats, scene->pool, scene->atlas, din.p, dout.p, n, scene->fast_ok ? 1 : 0);
  else
    hipLaunchKernelGGL((bounce_kernel<false, 1>), grid, block, 0, nullptr, scene->blob, scene->n_runs, scene->mats, scene->pool, scene->atlas, din.p, dout.p, n, scene->fast_ok ? 1 : 0);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpy(out, dout.p, (size_t)n * sizeof(PtBounceOut), hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_debug_schedule(const PtScene* scene, int32_t out[2]) {
  if (!scene || !out) return fail(PT_ERR_INVALID_ARG, "pt_debug_schedule: NULL argument");
  out[0] = out[1] = 0;
  PT_HIP(hipDeviceSynchronize());
  std::lock_guard<std::mutex> lock(scene->sched); // last_had_wide_phase / ws_nsplit belong to the launch path
  if (!scene->last_had_wide_phase) return PT_OK;
  int v[2] = {0, 0};
  PT_HIP(hipMemcpy(v, scene->ws_nsplit, sizeof v, hipMemcpyDeviceToHost));
  out[0] = v[0]; out[1] = v[0] > 0 ? (1 << v[1]) : 0;
  return PT_OK;
}

int pt_debug_last_launch(const PtScene* scene, int32_t out[4]) {
  if (!scene || !out) return fail(PT_ERR_INVALID_ARG, "pt_debug_last_launch: NULL argument");
  std::lock_guard<std::mutex> lock(scene->sched);
  for (int k = 0; k < 4; k++) out[k] = scene->last_launch[k];
  return PT_OK;
}

int pt_debug_camera_rays(const PtCamera* cam, int32_t width, int32_t height, const int32_t* xy, const uint32_t* rng_in,
                         PtCameraRay* out, int32_t n) {
  if (!cam || !xy || !rng_in || !out || n < 0 || width <= 0 || height <= 0)
    return fail(PT_ERR_INVALID_ARG, "pt_debug_camera_rays: bad argument");
  if (n == 0) return PT_OK;
  DevBuf<int32_t> dxy;
  DevBuf<uint32_t> drng;
  DevBuf<PtCameraRay> dout;
  PT_HIP(dxy.alloc((size_t)n * 2));
  PT_HIP(drng.alloc(n));
  PT_HIP(dout.alloc(n));
  PT_HIP(hipMemcpy(dxy.p, xy, (size_t)n * 2 * sizeof(int32_t), hipMemcpyHostToDevice));
  PT_HIP(hipMemcpy(drng.p, rng_in, (size_t)n * sizeof(uint32_t), hipMemcpyHostToDevice));
  Cam c;
  std::memcpy(&c, cam, sizeof c);
  hipLaunchKernelGGL(camera_rays_kernel, dim3((n + 63) / 64), dim3(64), 0, nullptr, c, width, height, dxy.p, drng.p, dout.p, n);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpy(out, dout.p, (size_t)n * sizeof(PtCameraRay), hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_debug_math(int32_t op, const float* a, const float* b, float* out, int64_t n) {
  if (!a || !out || n < 0 || op < 0 || op > 15) return fail(PT_ERR_INVALID_ARG, "pt_debug_math: bad argument");
  if ((op == 4 || op == 8 || op == 9 || op == 12 || op == 13) && !b) return fail(PT_ERR_INVALID_ARG, "pt_debug_math: op needs two operands");
  if (n == 0) return PT_OK;
  DevBuf<float> da, db, dout;
  PT_HIP(da.alloc(n));
  PT_HIP(dout.alloc(n));
  PT_HIP(hipMemcpy(da.p, a, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  if (b) {
    PT_HIP(db.alloc(n));
    PT_HIP(hipMemcpy(db.p, b, (size_t)n * sizeof(float), hipMemcpyHostToDevice));
  }
  hipLaunchKernelGGL(math_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, op, da.p, b ? db.p : nullptr, dout.p, (long long)n);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpy(out, dout.p, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
  return PT_OK;
}

int pt_debug_sphere_texel(const float* n_xyz, int64_t n, float freq, int32_t width, int32_t height, int32_t* out_ij, int32_t* exact_ij, uint8_t* took_fast,
                          float* uv4) {
  if (!n_xyz || !out_ij || !exact_ij || !took_fast || n < 0 || width < 1 || height < 1) return fail(PT_ERR_INVALID_ARG, "pt_debug_sphere_texel: bad argument");
  if (n == 0) return PT_OK;
  DevBuf<float> dn;
  DevBuf<int32_t> da, db;
  DevBuf<uint8_t> df;
  DevBuf<float> duv;
  if (uv4) PT_HIP(duv.alloc(4 * n));
  PT_HIP(dn.alloc(3 * n));
  PT_HIP(da.alloc(2 * n));
  PT_HIP(db.alloc(2 * n));
  PT_HIP(df.alloc(n));
  PT_HIP(hipMemcpy(dn.p, n_xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice));
  hipLaunchKernelGGL(sphere_texel_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, nullptr, dn.p, (long long)n, freq, (uint32_t)width, (uint32_t)height, da.p, db.p, df.p, uv4 ? duv.p : nullptr);
  PT_HIP(hipGetLastError());
  PT_HIP(hipMemcpy(out_ij, da.p, (size_t)n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
  PT_HIP(hipMemcpy(exact_ij, db.p, (size_t)n * 2 * sizeof(int32_t), hipMemcpyDeviceToHost));
  PT_HIP(hipMemcpy(took_fast, df.p, (size_t)n, hipMemcpyDeviceToHost));
  if (uv4) PT_HIP(hipMemcpy(uv4, duv.p, (size_t)n * 4 * sizeof(float), hipMemcpyDeviceToHost));
  return PT_OK;
}

} // extern "C"
