// Shared by sfe_cloudfilter.hip and sfe_store.hip: the per-frame header the resident cloud filters pass from kernel to
// kernel, and the octree root / depth of libpointmatcher's OctreeGridDataPointsFilter (Octree::build) from a bounding box.
#pragma once
#include "sfe_internal.h"

#define CF_MAX_LEVELS 31
#define CF_SORT_CAP 16384 // points per frame the LDS sort holds (128 KiB of 64-bit keys)
#define CF_MAX_CAP 65536  // 16 index bits in the sort key

struct CfHeader {
    float cx, cy, radius;
    int levels;
    int n;      // points of this frame (clamped to cap)
    int n_seg;  // leaves = points after the downsample
    int n_out;  // points after the outlier filter
    int zlev;   // >= 0: the downsampled cloud is in octree path order and its leaf keys (2 * zlev bits) were kept
};

// Octree::build: centre = min + radii*0.5, radius = max(radii)*0.5; split while a cell is wider than max_size
__device__ __forceinline__ CfHeader cf_make_header(float mnx, float mny, float mxx, float mxy, float max_size, int n)
{
    CfHeader h;
    const float rx = mxx - mnx, ry = mxy - mny;
    h.cx = mnx + rx * 0.5f;
    h.cy = mny + ry * 0.5f;
    float radius = rx;
    if (radius < ry)
        radius = ry;
    radius *= 0.5f;
    h.radius = radius;
    int L = 0;
    float r = radius;
    while (!((double)r * 2.0 <= (double)max_size) && L < CF_MAX_LEVELS) {
        r *= 0.5f;
        ++L;
    }
    h.levels = L;
    h.n = n;
    h.n_seg = n; // if the downsample is skipped the cloud passes through
    h.n_out = n;
    h.zlev = -1;
    return h;
}

// sfe_cloudfilter.hip: pcl.downsample (resolution > 0) and pcl.remove_outlier (min_points > 1) on n_frames clouds that
// a staging kernel has left as float2 in scratch slot 25 ([n_frames][cap]) with their headers in slot 27; outputs like
// sfe_cloud_filter_batch_dev.  Enqueue only.
int sfe_cf_run_staged(sfe_ctx *ctx, int n_frames, int64_t cap, float resolution, double radius, int min_points,
                      float *d_out, int32_t *d_out_counts);
// pcl.cpp:134 hands the resolution over as std::to_string(float): six decimals survive
float sfe_cf_max_size(float resolution);
#define CF_SLOT_P32 25
#define CF_SLOT_HDR 27
// Staged hand-over from the extraction (round 6): sfe_extract_points_bits_staged_dev leaves every frame's points as float2 in
// slot CF_SLOT_P32 -- the fp64 metres rounded to float32, the cast pybind makes at pcl.cpp's boundary -- and the frame's
// bounding box + point count here; sfe_cloud_filter_staged_dev turns the boxes into headers and runs the filters.  The
// float64 staging copy of rounds 2-5 (cf_cast_bbox_kernel: 16 B read + 8 B written per point) is gone from that path.
struct CfBBox {
    float mnx, mny, mxx, mxy; // over (float)y, (float)x of the frame's stored points (+-inf when it has none)
    int n;                    // stored points (the frame's count clamped to the capacity)
};
#define CF_SLOT_BBOX 63
