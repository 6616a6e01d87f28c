// pt_binned.hpp — the BINNED triangle-pool renderer (round 6): scenes with one pooled triangle run (BASELINE config 5: 100 k triangles).
// Included by pt_render.hip inside its anonymous namespace (it uses KArgs, LaneT and the lane_* functions of the persistent kernels).
//
// Why.  A pooled run's exact candidate set has two parts (pt_tripool.hpp): the grid — pairs a ray does not graze, found by a per-lane
// walk — and the direction map — the triangles a ray DOES graze, a list of 1 000 - 4 000 entries keyed by the ray's direction, of which
// ~55 survive the filters.  The persistent kernel took a wave's 64 rays through their 64 different lists one ray at a time, the list
// spread over the lanes: 2 300 gathered 16-byte records per ray, the kernel bound by its gathers (round 5: 9.4 s per frame, 21.7 TB past
// L2).  The list is a function of the direction BIN alone, so rays of one bin can share every record read — but the rays of one bin are
// spread over the whole frame.  This renderer therefore runs the frame as GENERATIONS: in one generation every live pixel traces ONE
// ray; its band stage is not run where the ray is, the rays are brought together by bin:
//
//   bin_step_kernel    one lane per pixel: finish the pixel's pending ray (the band stage's answer, the runs behind the pooled run,
//                      shading: the next ray or the next sample — the pixel's ONE generator stream, render.hpp:95-101, stays sequential
//                      because a pixel has one ray in flight), then start the next ray: the runs in front of the pooled run and the
//                      pool's grid part (tri_pool_scan<true>), and file a request under the ray's (rho class, direction bin) key
//   bin_count/offsets  a counting sort of the requests by key; the rays of a key are cut into PACKETS of up to 64
//   band_kernel        one wave per packet: 64 rays of ONE bin in the lanes, the bin's list streamed once — a record is read by one lane,
//                      staged in LDS and broadcast — each entry's filter (the same two necessary conditions as tri_pool_scan's band_pass,
//                      on the ready records of pt_tripool.hpp) evaluated for 64 rays per instruction; survivors run the reference's own
//                      test 64 pairs at a time; a hit lowers the ray's 64-bit slot in memory with the scan's own acceptance key.
//
// Exactness: every ray still meets exactly the candidates tri_pool_scan gives it — its grid walk, then every entry of its bin's list that
// passes the (relaxed, necessary) filter — and every candidate runs tri_param, the reference's test; the order in which candidates are
// tested does not matter (tri_key: smallest t, among equal t the last triangle in list order, triangle.hpp:91).  The other runs, the
// shading and the generator are the persistent kernels' own functions, called in the reference's order for every pixel.  The image is the
// persistent kernel's bit for bit (tests/test_gpu_parity.py: both renderers against the oracle and against each other).
#pragma once

#ifdef PT_BIN_DEBUG
__device__ unsigned int g_bin_fb[4096];
#endif
struct BinArgs {
  KArgs k; // MUST lead: lane_regenerate / lane_store read the camera and the frame geometry from the head of the kernarg segment
  f4* A0;  // per local pixel: ray origin, time
  f4* A1;  //                  ray direction, closest t of the pending ray so far
  f4* A2;  //                  attenuation, hit id of the pending ray so far
  f4* A3;  //                  radiance sum, generator state
  int4* A4; //                 samples done, bounces of this path, flags (1 live, 2 need a new sample), request key (< 0: none)
  f4* A5;  //                  request: rho, d.d, rank within its key
  unsigned long long* slot; // the pending ray's nearest hit as tri_key (band_kernel lowers it)
  unsigned int* hist;  // requests per key of this generation
  unsigned int* ctl;   // SortArgs::ctl: [3] live pixels of this generation (the length of live_in), [4] the count live_out grows by
  const unsigned int* live_in; // gen > 0: the live pixels, compacted by the previous generation (waves stay full as pixels finish)
  unsigned int* live_out;
  int pool_run;        // index of the pooled run in the run list
  int gen;
  int dbg_base1;       // (PT_BIN_DEBUG builds: the first key of the second map)
  int scatter_p;       // > 0: gen 0's thread -> pixel map is the stratified deal with this stride (coprime to the tile count)
};

__device__ __forceinline__ unsigned long long bin_key0(const HitState& h) {
  return h.hit >= 0 ? tri_key(h.closest, hit_off(h.hit)) : ((unsigned long long)0x7f800000u << 32);
}

template <int UV, int MATS>
__global__ __launch_bounds__(kBlock, PT_MIN_WAVES_BINSTEP) void bin_step_kernel(BinArgs ba) {
  const KArgs& a = ba.k;
  typedef LaneT<false> Lane;
  Lane L;
  lane_reset(L, (lds_fp) nullptr);
  const int t_ = (int)(blockIdx.x * kBlock + threadIdx.x);
  int n_here = a.n_local_pixels;
  if (ba.gen > 0) n_here = (int)((const __attribute__((address_space(4))) unsigned int*)(unsigned long long)ba.ctl)[3];
  if ((int)(blockIdx.x * kBlock) >= n_here) return;
  const bool in_range = t_ < n_here;
  // gen 0 deals the pixels to the threads as a stratified sample of the frame (lane_acquire's scatter: thread t takes pixel (t >> 6) + (t & 63) T'
  // ... of tile (t P) mod T): a generation lasts as long as its slowest wave, and a wave of one tile's 64 pixels walks 64 long or 64 short
  // rays; the live lists of the later generations inherit the mix
  int p = 0;
  if (ba.gen > 0) p = in_range ? (int)ba.live_in[t_] : 0;
  else if (ba.scatter_p > 0) {
    const unsigned int nt = (unsigned int)a.n_local_pixels >> 6, c = (unsigned int)t_;
    const unsigned int pos = c % nt, row = (c / nt) & 63u;
    p = (int)((((unsigned long long)pos * (unsigned int)ba.scatter_p) % nt) * 64u + row);
  } else p = t_;
  // local pixel -> frame pixel (lane_acquire's mapping: tiles dealt round-robin over shards, 64 pixels of a tile in a wave)
  const int l = p >> 6, in_tile = p & 63;
  const long long g = (long long)l * a.shard_count + a.shard_index;
  const int tx = (int)(g % a.tiles_x), ty = (int)(g / a.tiles_x);
  const int x = tx * PT_TILE + (in_tile & 7), y = ty * PT_TILE + (in_tile >> 3);
  const bool valid = in_range && g < a.n_tiles && x < a.width && y < a.height;
  int b_ = 0, s_ = 0;
  HitState h;
  hit_begin(h);
  if (ba.gen == 0) {
    L.live = valid;
    L.need_new = true;
    L.rng = (uint32_t)((unsigned long long)y * (unsigned long long)a.width + (unsigned long long)x); // render.hpp:130-132
    L.cold.begin(p, x, y);
  } else if (in_range) {
    const int4 m = ba.A4[p];
    L.live = (m.z & 1) != 0;
    L.need_new = (m.z & 2) != 0;
    s_ = m.x; b_ = m.y;
    L.cold.begin(p, x, y, s_);
    if (L.live) {
      const f4 a0 = ba.A0[p], a1 = ba.A1[p], a2 = ba.A2[p], a3 = ba.A3[p];
      L.ray.o = mk(a0.x, a0.y, a0.z); L.ray.tm = a0.w;
      L.ray.d = mk(a1.x, a1.y, a1.z); h.closest = a1.w;
      L.att = mk(a2.x, a2.y, a2.z); h.hit = as_i(a2.w);
      L.cold.resume(mk(a3.x, a3.y, a3.z), s_);
      L.rng = (uint32_t)as_i(a3.w);
      L.b = b_;
      // the band stage's answer: a triangle of the pooled run is the nearest hit so far iff the slot moved
      const unsigned long long kf = ba.slot[p];
      if (kf != bin_key0(h)) { h.closest = as_f((int)(unsigned int)(kf >> 32)); h.hit = hit_pack(DK_TRI, 0, (int)(0xffffffffu - (unsigned int)(kf & 0xffffffffull))); }
    }
  }
  const cst_f4p cblob = (cst_f4p)a.blob;
  // ---- finish the pending ray: the runs behind the pooled run, then emitted / scatter / sky and the sample bookkeeping -----------------
  if (ba.gen > 0 && __builtin_amdgcn_ballot_w64(L.live) != 0) {
    RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
    c.live = L.live;
    const bool fast = wave_all_regular(c, L.live);
    hit_world_range<UV == UV_TRACKED, false, 1, true, false, false>(cblob, cblob, ba.pool_run + 1, a.n_runs, c, fast, L.rng, h, a.pool);
    lane_shade<UV, false, MATS>(L, a, h, a.blob, a.mats, fast);
  }
  // ---- start the next ray: a new sample where the path ended, the runs up to and with the pooled run's grid part ------------------------
  TriDefer df;
  df.key = -1; df.rho = 0.0f;
  float ua = 0.0f;
#ifdef PT_BIN_DEBUG
  const unsigned long long dbg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
  if (__builtin_amdgcn_ballot_w64(L.live) != 0) {
    lane_regenerate(L, a);
    RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
    c.live = L.live;
    const bool fast = wave_all_regular(c, L.live);
    hit_begin(h);
    hit_world_range<UV == UV_TRACKED, false, 1, true, false, true>(cblob, cblob, 0, ba.pool_run + 1, c, fast, L.rng, h, a.pool, &df);
    ua = c.a;
  }
  {
    // the request's rank within its key: ONE atomic per distinct key of the wave (a tile's camera rays share a bin or two; 64 same-address
    // atomics per wave-instruction were what the first version of this kernel waited for)
    const int lane = threadIdx.x & 63;
    const unsigned long long below = (1ull << lane) - 1ull;
    const int key = (in_range && L.live) ? df.key : -1;
    unsigned long long peers = 0;
    for (unsigned long long todo = __builtin_amdgcn_ballot_w64(key >= 0); todo != 0;) {
      const int k0 = __builtin_amdgcn_readlane(key, __builtin_ctzll(todo));
      const unsigned long long m = __builtin_amdgcn_ballot_w64(key == k0);
      if (key == k0) peers = m;
      todo &= ~m;
    }
    unsigned int rank = 0;
    if (key >= 0) {
      const int leader = __builtin_ctzll(peers);
      unsigned int base = 0;
      if (lane == leader) base = atomicAdd(&ba.hist[key], (unsigned int)__builtin_popcountll(peers));
      rank = (unsigned int)__shfl((int)base, leader, 64) + (unsigned int)__builtin_popcountll(peers & below);
    }
    df.key = key;
    df.rho = key >= 0 ? df.rho : 0.0f;
    if (in_range && L.live) ba.A5[p] = f4{df.rho, ua, as_f((int)rank), 0.0f};
  }
  if (in_range) {
    const int key = df.key;
    if (L.live) {
      ba.A0[p] = f4{L.ray.o.x, L.ray.o.y, L.ray.o.z, L.ray.tm};
      ba.A1[p] = f4{L.ray.d.x, L.ray.d.y, L.ray.d.z, h.closest};
      ba.A2[p] = f4{L.att.x, L.att.y, L.att.z, as_f(h.hit)};
      const V3 acc = L.cold.get_acc();
      ba.A3[p] = f4{acc.x, acc.y, acc.z, as_f((int)L.rng)};
      ba.slot[p] = bin_key0(h);
    }
    ba.A4[p] = int4{L.cold.get_s(), L.b, (L.live ? 1 : 0) | (L.need_new ? 2 : 0), key};
  }
#ifdef PT_BIN_DEBUG /* ... and the longest phase 2 (the runs up to and with the pool's grid part) of a wave, in 10 ns ticks */
  {
    asm volatile("" ::"v"(h.closest), "v"(h.hit), "v"(df.key));
    const unsigned long long dbg_t1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) atomicMax(&g_bin_fb[1024 + min(ba.gen, 511)], (unsigned int)(dbg_t1 - dbg_t0));
  }
#endif
#ifdef PT_BIN_DEBUG /* diagnostic build: live rays per generation that filed no request (the pooled run was scanned in full by their wave) */
  {
    const unsigned long long fb = __builtin_amdgcn_ballot_w64(in_range && L.live && df.key < 0);
    if (fb != 0 && (threadIdx.x & 63) == 0) { atomicAdd(&g_bin_fb[2 * min(ba.gen, 511)], (unsigned int)__builtin_popcountll(fb)); atomicAdd(&g_bin_fb[2 * min(ba.gen, 511) + 1], 1u); }
  }
#endif
  // the pixels that are still live, compacted for the next generation
  const unsigned long long lv = __builtin_amdgcn_ballot_w64(in_range && L.live);
#ifdef PT_BIN_DEBUG /* ... live rays and requests by kind per generation */
  {
    const unsigned long long rq = __builtin_amdgcn_ballot_w64(in_range && L.live && df.key >= 0 && df.key < ba.dbg_base1);
    const unsigned long long rq2 = __builtin_amdgcn_ballot_w64(in_range && L.live && df.key >= ba.dbg_base1);
    if ((threadIdx.x & 63) == 0 && ba.gen < 512) { atomicAdd(&g_bin_fb[2048 + ba.gen], (unsigned int)__builtin_popcountll(lv)); atomicAdd(&g_bin_fb[2560 + ba.gen], (unsigned int)__builtin_popcountll(rq)); atomicAdd(&g_bin_fb[3072 + ba.gen], (unsigned int)__builtin_popcountll(rq2)); }
  }
#endif
  if (lv != 0) {
    const int lane = threadIdx.x & 63, leader = __builtin_ctzll(lv);
    unsigned int base = 0;
    if (lane == leader) base = atomicAdd(&ba.ctl[4], (unsigned int)__builtin_popcountll(lv));
    base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
    if (in_range && L.live) ba.live_out[base + (unsigned int)__builtin_popcountll(lv & ((1ull << lane) - 1ull))] = (unsigned int)p;
  }
}

// ---- the frame's tail: the pixels still live when generations stop paying, finished by persistent waves ---------------------------------
// A generation lasts as long as its slowest wave's walk, whatever the number of rays in it: once few pixels are live (the heaviest ones, some of
// them hundreds of generations from their end) a ray costs more here than in the persistent kernel, whose waves wait for nobody.  The launcher
// then hands the live pixels over (pt_render.hip: launch_binned): every lane of this kernel pulls a pixel from the last live list, picks its
// state up where the generations left it — the pending ray's nearest hit with the band stage's answer — and runs it to its last sample through
// the persistent kernel's own loop, the pooled run's direction-map part taken in place (tri_pool_scan<false>).
template <int UV, int MATS>
__global__ __launch_bounds__(kBlock, PT_MIN_WAVES_TRIPOOL) void bin_finish_kernel(BinArgs ba) {
  const KArgs& a = ba.k;
  typedef LaneT<false> Lane;
  Lane L;
  lane_reset(L, (lds_fp) nullptr);
  const int lane = threadIdx.x & 63;
  const cst_f4p cblob = (cst_f4p)a.blob;
  const unsigned int n_live = ((const __attribute__((address_space(4))) unsigned int*)(unsigned long long)ba.ctl)[3];
  // a wave's time is the SUM over its lanes (the band stage takes them one at a time): when there are fewer pixels than lanes every wave
  // takes its share (lane_acquire: lanes_cap)
  const unsigned int waves = gridDim.x * (kBlock / 64);
  const int cap = (int)min(64u, max(1u, (n_live + waves - 1u) / waves));
  bool pend = false; // this lane's pixel still has the ray pending that the generations left
  HitState hp;
  hit_begin(hp);
  for (;;) {
    // ---- idle lanes pull their next pixel (one atomic per wave)
    {
      const bool want = !L.live && !L.retired && lane < cap;
      const unsigned long long mask = __builtin_amdgcn_ballot_w64(want);
      if (mask != 0) {
        const int leader = __builtin_ctzll(mask);
        unsigned int base = 0;
        if (lane == leader) base = atomicAdd(&ba.ctl[5], (unsigned int)__builtin_popcountll(mask));
        base = (unsigned int)__builtin_amdgcn_readlane((int)base, leader);
        const unsigned int i = base + (unsigned int)__builtin_popcountll(mask & ((1ull << lane) - 1ull));
        if (want) {
          if (i >= n_live) L.retired = true;
          else {
            const int p = (int)ba.live_in[i];
            const int l = p >> 6, in_tile = p & 63;
            const long long g = (long long)l * a.shard_count + a.shard_index;
            const int tx = (int)(g % a.tiles_x), ty = (int)(g / a.tiles_x);
            const int4 m = ba.A4[p];
            const f4 a0 = ba.A0[p], a1 = ba.A1[p], a2 = ba.A2[p], a3 = ba.A3[p];
            L.cold.begin(p, tx * PT_TILE + (in_tile & 7), ty * PT_TILE + (in_tile >> 3), m.x);
            L.cold.resume(mk(a3.x, a3.y, a3.z), m.x);
            L.rng = (uint32_t)as_i(a3.w);
            L.b = m.y;
            L.need_new = (m.z & 2) != 0;
            L.ray.o = mk(a0.x, a0.y, a0.z); L.ray.tm = a0.w;
            L.ray.d = mk(a1.x, a1.y, a1.z); hp.closest = a1.w;
            L.att = mk(a2.x, a2.y, a2.z); hp.hit = as_i(a2.w);
            const unsigned long long kf = ba.slot[p];
            if (kf != bin_key0(hp)) { hp.closest = as_f((int)(unsigned int)(kf >> 32)); hp.hit = hit_pack(DK_TRI, 0, (int)(0xffffffffu - (unsigned int)(kf & 0xffffffffull))); }
            L.live = true;
            pend = true;
          }
        }
      }
    }
    if (__builtin_amdgcn_ballot_w64(L.live) == 0) {
      if (__builtin_amdgcn_ballot_w64(!L.retired && lane < cap) == 0) break;
      continue;
    }
    // ---- a resumed pixel's pending ray: the runs behind the pooled run, then the shading (the other lanes wait this short turn out)
    if (__builtin_amdgcn_ballot_w64(pend) != 0) {
      const bool others = L.live && !pend;
      L.live = pend;
      RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
      c.live = L.live;
      const bool fast = wave_all_regular(c, L.live);
      hit_world_range<UV == UV_TRACKED, false, 1, true, false, false>(cblob, cblob, ba.pool_run + 1, a.n_runs, c, fast, L.rng, hp, a.pool);
      lane_shade<UV, false, MATS>(L, a, hp, a.blob, a.mats, fast);
      L.live = L.live || others;
      pend = false;
      if (__builtin_amdgcn_ballot_w64(L.live) == 0) continue;
    }
    // ---- the persistent kernel's iteration
    lane_regenerate(L, a);
    RayCtx c = make_ctx(L.ray, a.fast_ok != 0);
    c.live = L.live;
    const bool fast = wave_all_regular(c, L.live);
    HitState h;
    hit_world<UV == UV_TRACKED, false, 1, true, false>(cblob, cblob, a.n_runs, c, fast, L.rng, h, a.pool);
    lane_shade<UV, false, MATS>(L, a, h, a.blob, a.mats, fast);
  }
}

// ---- the counting sort of a generation's requests -----------------------------------------------------------------------------------
// keys: [0, n_keys); hist[k] = requests under key k.  bin_count_kernel: per block of 1024 keys the number of requests and of packets
// (ceil(count / 64) each); bin_prefix_kernel: the exclusive prefix of the blocks' totals; bin_offsets_kernel: per key the offset of its
// rays in `sorted`, per packet its (key, chunk); zeroes the histogram for the next generation.
struct SortArgs {
  unsigned int* hist;    // [n_keys]
  unsigned int* offs;    // [n_keys + 1]
  uint2* block_tot;      // [n_blocks]: (requests, packets) of a block, then their exclusive prefix
  int2* packets;         // (key, chunk)
  unsigned int* ctl;     // [0] -, [1] n_packets, [2] band_kernel's packet counter, [3] live pixels of the generation (read by the host), [4] the accumulator bin_step_kernel adds to, [5] bin_finish_kernel's pixel queue
  int n_keys, n_blocks;
  int full_slices;       // a packet of the last key ("every triangle, exactly") is cut into this many slices of the run, one wave each
};

__global__ __launch_bounds__(1024) void bin_count_kernel(SortArgs sa) {
  __shared__ unsigned int wsum[16], wpk[16];
  const int k = (int)(blockIdx.x * 1024 + threadIdx.x);
  const unsigned int c = k < sa.n_keys ? sa.hist[k] : 0u;
  unsigned int s = c, q = ((c + 63u) >> 6) * (k == sa.n_keys - 1 ? (unsigned int)sa.full_slices : 1u);
#pragma unroll
  for (int st = 32; st >= 1; st >>= 1) { s += __shfl_xor(s, st, 64); q += __shfl_xor(q, st, 64); }
  if ((threadIdx.x & 63) == 0) { wsum[threadIdx.x >> 6] = s; wpk[threadIdx.x >> 6] = q; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned int ts = 0, tq = 0;
    for (int w = 0; w < 16; w++) { ts += wsum[w]; tq += wpk[w]; }
    sa.block_tot[blockIdx.x] = uint2{ts, tq};
  }
}

// exclusive prefix of the blocks' totals (a few hundred blocks: one wave, 64 at a time) and the generation's control words
__global__ __launch_bounds__(64) void bin_prefix_kernel(SortArgs sa) {
  unsigned int run_s = 0, run_q = 0;
  for (int base = 0; base < sa.n_blocks; base += 64) {
    const int i = base + (int)threadIdx.x;
    const uint2 v = i < sa.n_blocks ? sa.block_tot[i] : uint2{0u, 0u};
    unsigned int is = v.x, iq = v.y;
#pragma unroll
    for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o1 = __shfl_up(is, dd, 64), o2 = __shfl_up(iq, dd, 64); if ((int)threadIdx.x >= dd) { is += o1; iq += o2; } }
    if (i < sa.n_blocks) sa.block_tot[i] = uint2{run_s + is - v.x, run_q + iq - v.y};
    run_s += __shfl(is, 63, 64); run_q += __shfl(iq, 63, 64);
  }
  if (threadIdx.x == 0) {
    sa.offs[sa.n_keys] = run_s;
    sa.ctl[1] = run_q; sa.ctl[2] = 0;
    sa.ctl[3] = sa.ctl[4]; sa.ctl[4] = 0;
  }
}

__global__ __launch_bounds__(1024) void bin_offsets_kernel(SortArgs sa) {
  __shared__ unsigned int wsum[16], wpk[16];
  const int k = (int)(blockIdx.x * 1024 + threadIdx.x);
  const unsigned int c = k < sa.n_keys ? sa.hist[k] : 0u;
  const unsigned int slices = k == sa.n_keys - 1 ? (unsigned int)sa.full_slices : 1u;
  const unsigned int q = ((c + 63u) >> 6) * slices;
  unsigned int is = c, iq = q;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int dd = 1; dd < 64; dd <<= 1) { const unsigned int o1 = __shfl_up(is, dd, 64), o2 = __shfl_up(iq, dd, 64); if (lane >= dd) { is += o1; iq += o2; } }
  if (lane == 63) { wsum[w] = is; wpk[w] = iq; }
  __syncthreads();
  unsigned int bs = 0, bq = 0;
  for (int j = 0; j < w; j++) { bs += wsum[j]; bq += wpk[j]; }
  const uint2 bt = sa.block_tot[blockIdx.x];
  if (k < sa.n_keys) {
    const unsigned int off = bt.x + bs + is - c;
    unsigned int pk = bt.y + bq + iq - q;
    sa.offs[k] = off;
    for (unsigned int ch = 0; ch < q; ch++) sa.packets[pk + ch] = int2{k, (int)((ch / slices) | ((ch % slices) << 16))}; // (chunk | slice << 16)
    if (c) sa.hist[k] = 0u;
  }
}

__global__ __launch_bounds__(256) void bin_scatter_kernel(const int4* __restrict__ A4, const f4* __restrict__ A5, const unsigned int* __restrict__ offs,
                                                         unsigned int* __restrict__ sorted, const unsigned int* __restrict__ live, const unsigned int* __restrict__ ctl) {
  const unsigned int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= ctl[3]) return; // (bin_prefix_kernel has published this generation's live count)
  const unsigned int p = live[t];
  const int4 m = A4[p];
  if (m.w >= 0) sorted[offs[m.w] + (unsigned int)as_i(A5[p].z)] = p;
}

// ---- the band stage of a generation: one wave per packet ---------------------------------------------------------------------------
struct BandArgs {
  const f4* pool;
  const f4* blob;
  int hdr, goff;       // the pooled run's header in the blob; the offset of its first record (hit ids)
  const f4* A0;
  const f4* A1;
  const f4* A5;
  unsigned long long* slot;
  const unsigned int* offs;
  const int2* packets;
  unsigned int* ctl;   // SortArgs::ctl
  const unsigned int* sorted;
  int dense_min;       // packets of fewer rays take their rays one at a time (tri_band_one_ray)
  int full_slices;     // SortArgs::full_slices
};

constexpr int kBandWaves = 4; // waves per workgroup of band_kernel

__global__ __launch_bounds__(64 * kBandWaves, 4) void band_kernel(BandArgs b) {
#pragma clang fp contract(fast) /* filter arithmetic only (see tri_pool_scan): the reference's test — tri_param — is a function of its own, compiled as written */
  static_assert(kBandWaves * 64 == kBlock && PT_TRI_BQUEUE >= 64 * 3 && PT_TRI_QUEUE >= 128, "band_kernel borrows tri_pool_scan's per-wave LDS queues");
  __shared__ f4 rays[kBandWaves][64 * 2];
  __shared__ int pix[kBandWaves][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const cst_f4p cblob = (cst_f4p)b.blob;
  const glb_f4p pool = (glb_f4p)b.pool;
  const f4 H3 = cblob[b.hdr + 3], H4 = cblob[b.hdr + 4], H8 = cblob[b.hdr + 8], H12 = cblob[b.hdr + 12];
  const f4 D0 = cblob[b.hdr + 9], D1 = cblob[b.hdr + 10], D2 = cblob[b.hdr + 11];
  const int n_tri = as_i(H3.z), n_maps = as_i(H3.w);
  const unsigned int tri_sorted = (unsigned int)as_i(H4.z), ready = (unsigned int)as_i(H12.x);
  const int R0 = n_maps > 0 ? as_i(D0.x) : 0, R1 = n_maps > 1 ? as_i(D1.x) : 0, R2 = n_maps > 2 ? as_i(D2.x) : 0;
  const int base1 = 3 * R0 * R0, base2 = base1 + 3 * R1 * R1, base_all = base2 + 3 * R2 * R2;
  const unsigned long long below = (1ull << lane) - 1ull;
  f4* const st = tri_bqueue() + w * PT_TRI_BQUEUE; // the staged records of a trip (a dense packet), or tri_band_one_ray's stage-2 queue (a sparse one)
  f4* const ry = rays[w];
  int* const tq = tri_queue() + w * PT_TRI_QUEUE;
  int* const px = pix[w];
  const f4 H2 = cblob[b.hdr + 2], H5 = cblob[b.hdr + 5], H6 = cblob[b.hdr + 6], H7 = cblob[b.hdr + 7];
  TriBandCtx bctx;
  bctx.H5 = H5; bctx.H6 = H6; bctx.H7 = H7; bctx.H8 = H8; bctx.band_rec = (unsigned int)as_i(H4.w); bctx.tri_sorted = tri_sorted; bctx.n_tri = n_tri; bctx.goff = b.goff;
  const unsigned int n_packets = ((const __attribute__((address_space(4))) unsigned int*)(unsigned long long)b.ctl)[1];
  for (;;) {
    unsigned int pid = 0;
    if (lane == 0) pid = atomicAdd(&b.ctl[2], 1u);
    pid = (unsigned int)__builtin_amdgcn_readfirstlane((int)pid);
    if (pid >= n_packets) break;
    const int2 pk = b.packets[pid];
    const int key = __builtin_amdgcn_readfirstlane(pk.x), chunk = __builtin_amdgcn_readfirstlane(pk.y) & 0xffff, slice = __builtin_amdgcn_readfirstlane(pk.y) >> 16;
    const unsigned int o0 = b.offs[key], o1 = b.offs[key + 1];
    const int n = min(64, (int)(o1 - o0) - chunk * 64);
    // the list of this key: a bin of one of the maps, or every record
    unsigned int first = 0, last = (unsigned int)n_tri, cand_off = 0;
    const bool listed = key < base_all;
    if (listed) {
      const f4 D = key < base1 ? D0 : key < base2 ? D1 : D2;
      const unsigned int bin = (unsigned int)(key - (key < base1 ? 0 : key < base2 ? base1 : base2));
      const unsigned int foff = (unsigned int)as_i(D.z);
      first = sdword(pool, foff, bin); last = sdword(pool, foff, bin + 1u);
      cand_off = (unsigned int)as_i(D.w);
    }
    const bool on = lane < n;
    const unsigned int p = on ? b.sorted[o0 + (unsigned int)(chunk * 64 + lane)] : 0u;
    Ray r;
    float rho = 0.0f, ua = 1.0f;
    {
      const f4 a0 = b.A0[p], a1 = b.A1[p], a5 = b.A5[p];
      r.o = mk(a0.x, a0.y, a0.z); r.d = mk(a1.x, a1.y, a1.z); r.tm = 0.0f;
      rho = a5.x; ua = a5.y;
    }
    if (key > base_all) {
      // EVERY TRIANGLE, EXACTLY: rays outside the pool's domain (irregular, or too far out for the grid's slack) took no walk — the run is
      // scanned for them as the reference scans it, this wave's slice of it, a triangle broadcast to the packet's 64 rays at a time
      const int per = (n_tri + b.full_slices - 1) / b.full_slices;
      const int t0 = slice * per, t1 = min(n_tri, t0 + per);
      unsigned long long best = ~0ull;
      for (int base = t0; base < t1; base += 64) {
        const int cnt = min(64, t1 - base);
        const unsigned int o = tri_sorted + 3u * (unsigned int)(base + (lane < cnt ? lane : 0));
        const f4 T0 = pool[o], T1 = pool[o + 1], T2 = pool[o + 2];
        __builtin_amdgcn_wave_barrier();
        st[3 * lane] = T0; st[3 * lane + 1] = T1; st[3 * lane + 2] = T2;
        __builtin_amdgcn_wave_barrier();
        for (int j = 0; j < cnt; j++) {
          const f4 U0 = st[3 * j], U1 = st[3 * j + 1], U2 = st[3 * j + 2];
          float t;
          if (tri_param(U0, U1, U2, r, t) && !(t < PT_TMIN)) best = min(best, tri_key(t, b.goff + 3 * as_i(U2.w)));
        }
      }
      if (on && best != ~0ull) atomicMin(&b.slot[p], best);
      continue;
    }
    // per-ray constants of the filter (tri_pool_scan: rho and |d| rounded up)
    const float dn = __builtin_amdgcn_sqrtf(ua) * 1.000002f;
    if (n < b.dense_min) {
      // A SPARSE packet: too few rays to pay for streaming the list once per RAY-LANE; each ray in turn, the list's entries across the
      // lanes (tri_pool_scan's own routine: a gather per entry, but 64 entries per instruction)
      for (int i = 0; i < n; i++) {
        Ray ur;
        ur.o = mk(rl_f(r.o.x, i), rl_f(r.o.y, i), rl_f(r.o.z, i));
        ur.d = mk(rl_f(r.d.x, i), rl_f(r.d.y, i), rl_f(r.d.z, i));
        ur.tm = 0.0f;
        const unsigned int pi = (unsigned int)__builtin_amdgcn_readlane((int)p, i);
        tri_band_one_ray(pool, bctx, ur, rl_f(ua, i), rl_f(rho, i), rl_f(dn, i), first, last, cand_off, listed,
                         [&](unsigned long long key) { atomicMin(&b.slot[pi], key); });
      }
      continue;
    }
    __builtin_amdgcn_wave_barrier();
    ry[2 * lane] = f4{r.o.x, r.o.y, r.o.z, 0.0f};
    ry[2 * lane + 1] = f4{r.d.x, r.d.y, r.d.z, 0.0f};
    px[lane] = (int)p;
    const float dn1 = dn * 1.00001f, dne = dn * H8.w, rhodn = rho * dn * 1.001f, ua1 = ua * 1.00001f;
    int qn = 0;
    auto exact_batch = [&]() { // the reference's test for the top min(64, qn) queued (ray, triangle) pairs
      __builtin_amdgcn_wave_barrier();
      const int m = min(qn, 64);
      if (lane < m) {
        const unsigned int e = (unsigned int)tq[qn - m + lane];
        const int src = (int)(e >> 26);
        const f4 q0 = ry[2 * src], q1 = ry[2 * src + 1];
        Ray r2;
        r2.o = mk(q0.x, q0.y, q0.z); r2.d = mk(q1.x, q1.y, q1.z); r2.tm = 0.0f;
        const unsigned int o = tri_sorted + 3u * (e & 0x3ffffffu);
        const f4 T0 = pool[o], T1 = pool[o + 1], T2 = pool[o + 2];
        float t;
        if (tri_param(T0, T1, T2, r2, t) && !(t < PT_TMIN)) atomicMin(&b.slot[px[src]], tri_key(t, b.goff + 3 * as_i(T2.w)));
      }
      qn -= m;
      __builtin_amdgcn_wave_barrier();
    };
    for (unsigned int base = first; base < last; base += 64u) {
      const int cnt = (int)min(64u, last - base);
      unsigned int idx = 0;
      if (lane < cnt) idx = listed ? gdword_stream(pool, cand_off, base + (unsigned int)lane) : base + (unsigned int)lane;
      idx = min(idx, (unsigned int)n_tri + 63u);
      {
        const unsigned int o = ready + 3u * idx;
        const f4 Q0 = pool[o], Q1 = pool[o + 1], Q2 = pool[o + 2];
        __builtin_amdgcn_wave_barrier();
        st[3 * lane] = Q0; st[3 * lane + 1] = Q1; st[3 * lane + 2] = Q2;
        __builtin_amdgcn_wave_barrier();
      }
      for (int j = 0; j < cnt; j++) {
        const f4 Q0 = st[3 * j], Q1 = st[3 * j + 1], Q2 = st[3 * j + 2]; // (n~ x 32767, pn~) (C~, Lr) (G, nlow, E2, KR): pt_tripool.hpp "ready records"
        const float dq = __builtin_fabsf(r.d.x * Q0.x + r.d.y * Q0.y + r.d.z * Q0.z) * 3.0518509e-5f; // |d . n~|
        const bool band = dq <= dn1 * (Q0.w * rho + Q2.x);
        if (__builtin_amdgcn_ballot_w64(band && on) == 0) continue;
        const float a1 = (dq - dne) * Q2.y - Q2.z * dn;                          // <= |a'| - ea |d|
        const float rad = Q1.w + Q2.w * rhodn * __builtin_amdgcn_rcpf(a1);       // >= L + the noise radius (a1 <= 0: no bound)
        const V3 xx = cross(mk(Q1.x, Q1.y, Q1.z) - r.o, r.d);
        const bool pass = on && band && (!(a1 > 0.0f) || dot(xx, xx) <= rad * rad * ua1);
        const unsigned long long m = __builtin_amdgcn_ballot_w64(pass);
        if (m != 0) {
          const unsigned int e = (unsigned int)__builtin_amdgcn_readlane((int)idx, j);
          if (pass) tq[qn + __builtin_popcountll(m & below)] = (int)(((unsigned int)lane << 26) | e);
          qn += __builtin_popcountll(m);
          if (qn >= 64) exact_batch();
        }
      }
    }
    while (qn > 0) exact_batch();
  }
}
