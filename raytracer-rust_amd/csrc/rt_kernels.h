// Launch wrappers of the gfx950 kernels (rt_kernels.hip).
#pragma once
#include <algorithm>

#include "rt_device.h"

namespace rtamd {

struct LaunchCfg {
    uint32_t n_cu = 256;
};

void launch_raygen(const LaunchCfg& cfg, const DevScene& sc, const PassParams& pp, RayQueue q, float4* radiance, uint32_t* ctl,
                   uint32_t capacity, hipStream_t st);
void launch_raygen_entry(const LaunchCfg& cfg, const DevScene& sc, const PassParams& pp, RayQueue q, float4* radiance, uint32_t* ctl, int4* table,
                         hipStream_t st);   // k_raygen and k_tile_entry in one launch
void launch_pass_begin(uint32_t* ctl, uint32_t n_paths, hipStream_t st);
void launch_tile_entry(const DevScene& sc, const PassParams& pp, int4* table, hipStream_t st);
void launch_closest_hit(const LaunchCfg& cfg, const DevScene& sc, RayQueue q, float2* hits, uint32_t* ctl, int it, uint32_t capacity,
                        uint32_t max_rays, bool counters, TileEntry entry, bool any_hit, const PassParams* gen, hipStream_t st);   // gen != NULL: iteration 0 forms its rays itself (no k_raygen)
void launch_shade(const LaunchCfg& cfg, const DevScene& sc, RayQueue qi, const float2* hits, RayQueue qo, float4* radiance,
                  uint32_t* ctl, int it, int last, uint32_t capacity, uint32_t max_rays, const PassParams* gen, int cls, uint32_t* heavy, hipStream_t st);   // cls: k_shade's CLS (0: one launch; 1, 2: the two launches of a split iteration)
void launch_accumulate(const DevScene& sc, const PassParams& pp, const float4* radiance, float* accum, hipStream_t st);
void launch_resolve(const float* accum, size_t n_values, float inv_spp, uint8_t* rgb8, float* color, hipStream_t st);
void launch_hit_info(const DevScene& sc, const PassParams& pp, RayQueue q, const float2* hits, const uint32_t* ctl, uint32_t capacity,
                     int32_t* obj, int32_t* tri, float* t, float* nrm, float* dir, hipStream_t st);
void launch_accum_add(const LaunchCfg& cfg, float* dst, const float* src, size_t n, hipStream_t st);
void launch_calib_gather(const float4* table, uint32_t n_records, uint32_t iters, uint32_t blocks, float* sink, hipStream_t st);
void launch_stats(const uint32_t* ctl, unsigned long long* stats, uint32_t capacity, hipStream_t st);

}  // namespace rtamd
