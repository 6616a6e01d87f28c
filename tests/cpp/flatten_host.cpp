// Host-only translation unit for tests/test_flatten_sanitize.py: the product's scene flattener (pt_flatten.hpp + pt_tripool.hpp — grid
// builders, slab pools, fine grids with neighbour bits, direction maps, Morton copies, quantised records: ~1 000 lines of raw offsets) compiled with
// g++ -fsanitize=address,undefined and driven from Python through ctypes.  No HIP, no GPU.  The reference's analogue is its sanitizer
// build options (CMakeLists.txt:76-90).
#include <cstdint>
#include <cstring>
#include <string>

#include "../../path_tracer_amd/csrc/pt_flatten.hpp"

// pool_f4 / pool_out: the triangle pools' tables (a buffer of their own since round 5: pt_flatten.hpp PoolLayout), assembled as pt_scene_create
// uploads them; every segment is read once more for the checksum (the big maps are checksummed where they lie, not copied)
extern "C" int flat_check(const PtSceneDesc* desc, int allow_grid, int box_cull, int allow_tri, int tri_min_run, int64_t* blob_f4,
                          uint64_t* checksum, int32_t* stats, float* blob_out, int64_t blob_cap_f4, int64_t* pool_f4, float* pool_out, int64_t pool_cap_f4) {
  ptf::Flat flat;
  std::string err;
  ptf::TriPoolTuning tri;
  if (tri_min_run > 0) tri.min_run = tri_min_run;
  const int rc = ptf::flatten(desc, flat, err, allow_grid != 0, box_cull, ptf::GridTuning(), allow_tri != 0, tri);
  if (rc) return rc;
  uint64_t h = 1469598103934665603ull; // FNV-1a over the blob and the material table: every byte is read once more under ASan
  const unsigned char* p = reinterpret_cast<const unsigned char*>(flat.blob.data());
  for (size_t i = 0; i < flat.blob.size() * 16; i++) h = (h ^ p[i]) * 1099511628211ull;
  p = reinterpret_cast<const unsigned char*>(flat.mats.data());
  for (size_t i = 0; i < flat.mats.size() * 16; i++) h = (h ^ p[i]) * 1099511628211ull;
  for (const ptf::PoolSegment& sg : flat.pool.segments) {
    p = reinterpret_cast<const unsigned char*>(sg.dwords.data());
    for (size_t i = 0; i < sg.dwords.size() * 4; i += 64) h = (h ^ p[i]) * 1099511628211ull; // (a byte per cache line: the maps are hundreds of MB)
  }
  if (pool_f4) *pool_f4 = (int64_t)flat.pool.size_f4;
  if (pool_out && pool_cap_f4 >= (int64_t)flat.pool.size_f4 && flat.pool.size_f4) flat.pool.assemble(reinterpret_cast<ptf::F4*>(pool_out));
  if (blob_f4) *blob_f4 = (int64_t)flat.blob.size();
  if (checksum) *checksum = h;
  if (stats) { stats[0] = flat.n_runs; stats[1] = flat.grid_spheres; stats[2] = flat.pooled; stats[3] = flat.tri_pooled; }
  if (blob_out && !flat.blob.empty() && blob_cap_f4 >= (int64_t)flat.blob.size()) std::memcpy(blob_out, flat.blob.data(), flat.blob.size() * 16);
  return 0;
}
