// Level table of the multiresolution hash encoding (BASELINE config 5; oracle/hashgrid_oracle.py cites the definition).
#pragma once
#include <stdint.h>

namespace snr {

constexpr int kHgLevels = 16;
// sigma_net 32x64 + 64x16, color_net 32x64 + 64x64 + 64x16 (rows [out][in], run_nerf_helpers_tcnn.py:48-84)
constexpr int kHgNetParams = 2048 + 1024 + 2048 + 4096 + 1024;

struct HgLevel {
  float scale;        // base_resolution * per_level_scale^l - 1
  uint32_t res;       // ceil(scale) + 1
  uint32_t size;      // entries of this level (dense: res^3 rounded up to 8; else 2^19)
  uint32_t offset;    // first entry
  uint32_t hashed;
};
struct HgTable {
  HgLevel level[kHgLevels];
  uint32_t entries;
  float bound, inv_2bound;
};

}  // namespace snr
