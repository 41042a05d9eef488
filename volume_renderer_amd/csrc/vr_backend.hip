// vr_backend.hip — C-ABI implementation (include/vr_backend.h) over the HIP kernels.
//
// Host side of the drop-in boundary: copies the scene into HBM (bricked density + skip grid,
// threaded BVH), restates the per-frame camera constants of BaseIntegrator::renderPartial, and
// launches the integrator kernels in sample passes that fit the staging buffer.  Nothing here
// touches oracle/ — if the GPU or the HIP runtime is missing every entry point fails with
// VR_ERR_NO_DEVICE / VR_ERR_HIP; there is no CPU fallback.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/vr_backend.h"
#include "vr_kernels.h"

using namespace vr;

#define VR_MAJOR_LEVELS 5   // majorant cells of 8^3 (the bricks) ... 128^3 voxels
#define VR_SKIP_CAP 31   // largest empty-cube radius the skip grid records, in voxels

namespace {

thread_local std::string g_create_error;

// ---------------------------------------------------------------------------------------------
// set-up kernels
// ---------------------------------------------------------------------------------------------

// dense (k fastest) -> 8^3 bricks, padding with `background`
__global__ void __launch_bounds__(256) k_rebrick(const float* __restrict__ dense, float* __restrict__ bricks, int dx, int dy,
                                                 int dz, int nbx, int nby, int nbz, float background)
{
    const size_t total = (size_t)nbx * nby * nbz * 512;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const uint32_t local = (uint32_t)(o & 511);
        size_t         b = o >> 9;
        const int bk = (int)(b % nbz); b /= nbz;
        const int bj = (int)(b % nby);
        const int bi = (int)(b / nby);
        int li, lj, lk;
        brick_local_decode(local, li, lj, lk);
        const int i = bi * 8 + li, j = bj * 8 + lj, k = bk * 8 + lk;
        float     v = background;
        if (i < dx && j < dy && k < dz) v = dense[((size_t)i * dy + j) * dz + k];
        bricks[o] = v;
    }
}
__global__ void __launch_bounds__(256) k_unbrick(const float* __restrict__ bricks, float* __restrict__ dense, int dx, int dy,
                                                 int dz, int nby, int nbz)
{
    const size_t total = (size_t)dx * dy * dz;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const int    k = (int)(o % dz);
        const size_t q = o / dz;
        const int    j = (int)(q % dy), i = (int)(q / dy);
        const size_t b = ((size_t)(i >> 3) * nby + (j >> 3)) * nbz + (k >> 3);
        dense[o] = bricks[(b << 9) | brick_local((uint32_t)i, (uint32_t)j, (uint32_t)k)];
    }
}
// one wave per brick: majorant[b] = max over the brick's 512 voxels
__global__ void __launch_bounds__(256) k_majorant(const float* __restrict__ bricks, float* __restrict__ majorant, uint32_t nbricks)
{
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int      lane = threadIdx.x & 63;
    if (wave >= nbricks) return;
    const float4* src = reinterpret_cast<const float4*>(bricks + ((size_t)wave << 9)) + lane * 2;
    const float4  a = src[0], b = src[1];
    float m = fmaxf(fmaxf(fmaxf(a.x, a.y), fmaxf(a.z, a.w)), fmaxf(fmaxf(b.x, b.y), fmaxf(b.z, b.w)));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) majorant[wave] = m;
}

// ---- skip grid ----------------------------------------------------------------------------
// For every empty voxel (value <= 0) the L-infinity distance d >= 1, in voxels and capped, to the
// nearest voxel that is not empty or lies outside the bbox: every voxel of the cube of radius
// d - 1 around it is then empty too, and a walk can take all its empty-space steps up to that
// cube's boundary at once (empty_run).  The distance is separable: along k it is a 1-D distance,
// along j then  d2(j) = min_j' max(|j - j'|, d1(j')),  along i the same over d2.
__device__ __forceinline__ size_t brick_index(int i, int j, int k, int nby, int nbz)
{
    return ((((size_t)(i >> 3) * nby + (j >> 3)) * nbz + (k >> 3)) << 9) | (size_t)brick_local((uint32_t)i, (uint32_t)j, (uint32_t)k);
}
__global__ void __launch_bounds__(256) k_skip_pass(const float* __restrict__ bricks, const uint8_t* __restrict__ in,
                                                   uint8_t* __restrict__ out, int dx, int dy, int dz, int nbx, int nby, int nbz,
                                                   int axis, int cap)
{
    const size_t total = (size_t)nbx * nby * nbz * 512;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const uint32_t local = (uint32_t)(o & 511);
        size_t         b = o >> 9;
        const int bk = (int)(b % nbz); b /= nbz;
        const int bj = (int)(b % nby);
        const int bi = (int)(b / nby);
        int li, lj, lk;
        brick_local_decode(local, li, lj, lk);
        const int i = bi * 8 + li, j = bj * 8 + lj, k = bk * 8 + lk;
        if (i >= dx || j >= dy || k >= dz) { out[o] = 0; continue; }
        const int c = axis == 2 ? k : (axis == 1 ? j : i), n = axis == 2 ? dz : (axis == 1 ? dy : dx);
        auto at = [&](int cc) -> int {
            if (cc < 0 || cc >= n) return 0;
            const size_t oo = axis == 2 ? brick_index(i, j, cc, nby, nbz) : (axis == 1 ? brick_index(i, cc, k, nby, nbz) : brick_index(cc, j, k, nby, nbz));
            return axis == 2 ? (bricks[oo] > 0.0f ? 0 : cap + 1) : (int)in[oo];
        };
        int best = at(c);
        for (int d = 1; d < best; ++d) {
            const int a = max(d, at(c - d)), e = max(d, at(c + d));
            best = min(best, min(a, e));
        }
        out[o] = (uint8_t)min(best, cap + 1);
    }
}
// skip[v] = value where it is > 0, else -(distance): sigma = multiplier * skip[v] keeps its sign for
// multipliers >= 0, which is all the walks read of an empty voxel
__global__ void __launch_bounds__(256) k_skip_make(const float* __restrict__ bricks, const uint8_t* __restrict__ dist,
                                                   float* __restrict__ skip, size_t total)
{
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const float v = bricks[o];
        const int   d = dist[o];
        skip[o] = (v > 0.0f || d == 0) ? v : -(float)d;     // d == 0: padding outside the bbox, never looked up
    }
}

// the synthetic cloud of BASELINE configs 2-4, written straight into brick order
// (same expression sequence as the oracle's cloud_density)
__global__ void __launch_bounds__(256) k_cloud(float* __restrict__ bricks, int n, int nb, float R)
{
    const size_t total = (size_t)nb * nb * nb * 512;
    for (size_t o = (size_t)blockIdx.x * blockDim.x + threadIdx.x; o < total; o += (size_t)gridDim.x * blockDim.x) {
        const uint32_t local = (uint32_t)(o & 511);
        size_t         b = o >> 9;
        const int bk = (int)(b % nb); b /= nb;
        const int bj = (int)(b % nb);
        const int bi = (int)(b / nb);
        int li, lj, lk;
        brick_local_decode(local, li, lj, lk);
        const int i = bi * 8 + li, j = bj * 8 + lj, k = bk * 8 + lk;
        const float half = 0.5f * (float)n;
        const float px = ((float)i + 0.5f) - half, py = ((float)j + 0.5f) - half, pz = ((float)k + 0.5f) - half;
        const float r = sqrtf(px * px + py * py + pz * pz);
        float t = 1.0f - r / R;
        t = t < 0.0f ? 0.0f : (t > 1.0f ? 1.0f : t);
        const float fall = t * t * (3.0f - 2.0f * t);
        float d = 0.0f;
        if (fall > 0.0f) {
            float f = 4.0f / (float)n, amp = 1.0f, sum = 0.0f;
            for (int oc = 0; oc < 4; ++oc) {
                sum = sum + amp * vr::perlin_noise(px * f + 17.0f, py * f + 17.0f, pz * f + 17.0f);
                f = f * 2.0f;
                amp = amp * 0.5f;
            }
            d = fall * (sum > 0.0f ? sum : 0.0f);
        }
        bricks[o] = d;
    }
}

// max channel of an RGB buffer (BaseIntegrator.cpp:199-209): non-negative floats compare like ints
__global__ void __launch_bounds__(256) k_max_channel(const float* __restrict__ rgb, size_t n, int* __restrict__ out_bits)
{
    float m = 0.0f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float v = rgb[i];
        if (v > m) m = v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((threadIdx.x & 63) == 0) atomicMax(out_bits, __float_as_int(m));
}
// extended Reinhard (BaseIntegrator.cpp:211-220)
__global__ void __launch_bounds__(256) k_tonemap(float* __restrict__ rgb, size_t n, float max_white2)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        float p = rgb[i];
        float numerator = p * ((p / max_white2) + 1.0f);
        rgb[i] = numerator / (p + 1.0f);
    }
}

// BaseIntegrator::assignPointToQuadrant (BaseIntegrator.cpp:112-129) for every sub-sample of a frame, on the
// device: the recursion unrolled into a descent (which quadrant at each level) and the sums taken from the
// innermost level outwards, as the recursion returns them.  spp <= 2^20: at most nine levels above the base case.
__global__ void __launch_bounds__(256) k_quadrant_offsets(float2* __restrict__ out, uint32_t spp)
{
    for (uint32_t s = blockIdx.x * blockDim.x + threadIdx.x; s < spp; s += gridDim.x * blockDim.x) {
        int idx[12], tot[12], n = 0;
        int i = (int)s, total = (int)spp;
        while (total > 4 && n < 12) {                               // :123 recurse with (i % (total/4), total/4)
            idx[n] = i; tot[n] = total; n++;
            i = i % (total / 4); total = total / 4;
        }
        float ox = 0.0f, oy = 0.0f;                                 // :113 total < 4
        if (total == 4) {                                           // :115-120
            ox = (i == 0 || i == 1) ? 0.25f : -0.25f;
            oy = (i == 0 || i == 2) ? 0.25f : -0.25f;
        }
        while (n-- > 0) {
            i = idx[n]; total = tot[n];
            const float value = 1.0f / total;                       // :122
            float sx, sy;
            if (i < total / 4.0f)          { sx = value;  sy = value; }
            else if (i < 2 * total / 4.0f) { sx = value;  sy = -value; }
            else if (i < 3 * total / 4.0f) { sx = -value; sy = value; }
            else                           { sx = -value; sy = -value; }
            ox = sx + ox; oy = sy + oy;                             // :128 offset + recursive result
        }
        out[s] = make_float2(ox, oy);
    }
}

// the event phases' parameter block goes to HBM as a kernel argument: no host buffer has to outlive the call
__global__ void k_store_cold(const TrackCold c, TrackCold* __restrict__ dst)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) *dst = c;
}

// ---------------------------------------------------------------------------------------------
// scratch of the allocation-free entry point (vr_backend_render_row_set_device_noalloc)
// ---------------------------------------------------------------------------------------------
// One array per GPU for the life of the process: work counters, statistics counters, the cold parameter block,
// the sub-sample offsets and the per-sample staging buffer of ONE launch.  In host code the name is a stand-in,
// not an address: vr_backend_create takes its address on the backend's device with hipGetSymbolAddress and
// everything after that uses the pointer.
#define VR_NOALLOC_HEADER_BYTES  (1u << 20)                     // counters, cold block, offsets of <= 65 536 sub-samples
#define VR_NOALLOC_SAMPLE_BYTES  (128ull << 20)                 // staging of one sample pass: <= 11 184 810 pixels per row set
#define VR_NOALLOC_MAX_SPP       65536u
#define VR_NOALLOC_OFF_WORK      0u
#define VR_NOALLOC_OFF_COUNTERS  256u
#define VR_NOALLOC_OFF_COLD      1024u
#define VR_NOALLOC_OFF_OFFSETS   (512u << 10)
static __device__ __align__(256) unsigned char scratch[VR_NOALLOC_HEADER_BYTES + VR_NOALLOC_SAMPLE_BYTES];
static_assert(VR_NOALLOC_OFF_COLD + sizeof(TrackCold) <= VR_NOALLOC_OFF_OFFSETS, "cold block overlaps the offsets");
static_assert(VR_NOALLOC_OFF_OFFSETS + VR_NOALLOC_MAX_SPP * sizeof(float2) <= VR_NOALLOC_HEADER_BYTES, "offsets overlap the samples");
static_assert(sizeof(Counters) <= VR_NOALLOC_OFF_COLD - VR_NOALLOC_OFF_COUNTERS, "counters overlap the cold block");

struct Pass { hipEvent_t e0, e1; };

} // namespace

struct vr_backend {
    int           device = 0;
    hipStream_t   stream = nullptr;
    std::string   err;
    int           num_cu = 256;
    int           mode = VR_MODE_FAST;
    uint64_t      seed = 0x5EED0001ull;
    uint64_t      sample_budget = 32ull << 30;   // of 288 GB: config 4 (4K at 256 spp, 25.5 GB of samples) in one launch

    // density / temperature
    bool     has_density = false, has_temperature = false;
    float*   d_bricks = nullptr;  float* d_skip = nullptr;  GridDev grid{};   // d_skip: see k_skip_pass
    float*   d_major = nullptr;                                                // per-brick maximum, then the same pooled over 2^3 ... 16^3 bricks (vr_delta_dda_kernel)
    size_t   major_off[VR_MAJOR_LEVELS] = {}; int32_t major_nc[VR_MAJOR_LEVELS][3] = {};          // offset and cell counts of each level
    float*   d_tbricks = nullptr; GridDev tgrid{};
    float    inv_mat[9]{}, vec[3]{}, mat[9]{};
    float    bmin[3]{}, bmax1[3]{};
    float    max_value = 0.0f;

    // mesh
    bool      has_mesh = false;
    BvhNode*  d_nodes = nullptr; TriDev* d_tris = nullptr; uint32_t* d_p2s = nullptr; float* d_dissolve = nullptr; float* d_diffuse = nullptr;
    uint32_t  nnodes = 0, nshapes = 0;

    // scratch
    float*    d_samples = nullptr; size_t samples_cap = 0;
    float*    d_rows = nullptr;    size_t rows_cap = 0;
    float2*   d_offsets = nullptr; size_t offsets_cap = 0;
    uint32_t* d_work = nullptr;
    Counters* d_counters = nullptr;
    int*      d_maxbits = nullptr;

    TrackCold*        d_cold = nullptr;
    std::vector<Pass> passes;
    size_t            passes_used = 0;
    uint32_t          launches = 0;           // integrator kernel launches of the last render call
    vr_render_stats   stats{};
    Counters*         last_counters = nullptr; // where the last render call counted (d_counters or the no-alloc scratch)

    // this device's copy of `scratch` (hipGetSymbolAddress at create)
    unsigned char*    d_noalloc = nullptr;

    // developer knobs (VR_TUNE_*), read at the start of each render call; 0 = built-in default
    int  tune_no_geom = 0, tune_no_lds_major = 0;
    bool last_geom = false;          // the last tracking launch took the closed-form voxel runs (GEOM)
    int tune_tile_h_log = 0, tune_round_max = 0, tune_round_events = 0, tune_chunk = 0, tune_march_tile_h_log = 0, tune_bvh_leaf = 0, tune_major_level = 0, tune_bvh_median = 0;
    // hipOccupancyMaxActiveBlocksPerMultiprocessor per kernel, asked once
    struct { const void* kernel; int blocks; } occupancy[64]{};
    int noccupancy = 0;
};

namespace {

int fail(vr_backend* b, int code, const char* fmt, ...)
{
    char    buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (b) b->err = buf; else g_create_error = buf;
    return code;
}
#define HIP_TRY(b, call)                                                                              \
    do {                                                                                              \
        hipError_t e_ = (call);                                                                       \
        if (e_ != hipSuccess)                                                                         \
            return fail((b), e_ == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP, "%s: %s", #call, \
                        hipGetErrorString(e_));                                                       \
    } while (0)

int use_device(vr_backend* b)
{
    HIP_TRY(b, hipSetDevice(b->device));
    return VR_OK;
}

template <typename T>
int ensure(vr_backend* b, T*& ptr, size_t& cap, size_t count)
{
    if (count <= cap && ptr) return VR_OK;
    T* fresh = nullptr;     // the old buffer stays until the larger one exists: a failed grow leaves it usable
    HIP_TRY(b, hipMalloc((void**)&fresh, count * sizeof(T)));
    if (ptr) (void)hipFree(ptr);
    ptr = fresh; cap = count;
    return VR_OK;
}

int grid_blocks(size_t n, int cap = 8192)
{
    size_t g = (n + 255) / 256;
    return (int)std::max<size_t>(1, std::min<size_t>(g, (size_t)cap));
}

int upload_grid(vr_backend* b, const vr_grid_desc* g, float*& d_bricks, GridDev& out)
{
    int64_t d[3];
    for (int i = 0; i < 3; ++i) {
        d[i] = (int64_t)g->bbox_max[i] - g->bbox_min[i] + 1;
        if (d[i] <= 0 || d[i] > 4096) return fail(b, VR_ERR_INVALID_ARG, "grid extent %lld on axis %d out of range", (long long)d[i], i);
    }
    if (!g->voxels) return fail(b, VR_ERR_INVALID_ARG, "grid voxels == NULL");
    const size_t nvox = (size_t)d[0] * d[1] * d[2];
    const int    nb[3] = { (int)((d[0] + 7) / 8), (int)((d[1] + 7) / 8), (int)((d[2] + 7) / 8) };
    const size_t nbricks = (size_t)nb[0] * nb[1] * nb[2];
    if ((nbricks << 9) >= (1ull << 32)) return fail(b, VR_ERR_INVALID_ARG, "grid too large for 32-bit voxel offsets");
    float* d_dense = nullptr;
    HIP_TRY(b, hipMalloc((void**)&d_dense, nvox * sizeof(float)));
    hipError_t e = hipMemcpyAsync(d_dense, g->voxels, nvox * sizeof(float), hipMemcpyHostToDevice, b->stream);
    if (e == hipSuccess) {
        if (d_bricks) { (void)hipFree(d_bricks); d_bricks = nullptr; }
        e = hipMalloc((void**)&d_bricks, nbricks * 512 * sizeof(float));
    }
    if (e == hipSuccess) {
        k_rebrick<<<grid_blocks(nbricks * 512), 256, 0, b->stream>>>(d_dense, d_bricks, (int)d[0], (int)d[1], (int)d[2], nb[0], nb[1],
                                                                      nb[2], g->background);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(d_dense);
    if (e != hipSuccess) return fail(b, e == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP, "grid upload: %s", hipGetErrorString(e));
    out.bricks = d_bricks;
    for (int i = 0; i < 3; ++i) { out.min[i] = g->bbox_min[i]; out.dim[i] = (int32_t)d[i]; out.nb[i] = nb[i]; }
    out.background = g->background;
    return VR_OK;
}

// max value (nanovdb::getExtrema(grid, bbox).max(), SceneLoader.cpp:219) and the skip grid
int build_derived(vr_backend* b)
{
    const GridDev& g = b->grid;
    const uint32_t nbricks = (uint32_t)g.nb[0] * g.nb[1] * g.nb[2];
    const size_t   total = (size_t)nbricks * 512;
    if (b->d_major) { (void)hipFree(b->d_major); b->d_major = nullptr; }
    // kept: the coarse majorant grids of the local-majorant delta tracker (level 0 = bricks, levels 1, 2 pooled below)
    size_t major_total = 0;
    for (int l = 0; l < VR_MAJOR_LEVELS; ++l) {
        b->major_off[l] = major_total;
        for (int a = 0; a < 3; ++a) b->major_nc[l][a] = (g.nb[a] + (1 << l) - 1) >> l;
        major_total += (size_t)b->major_nc[l][0] * b->major_nc[l][1] * b->major_nc[l][2];
    }
    HIP_TRY(b, hipMalloc((void**)&b->d_major, major_total * sizeof(float)));
    float* const d_major = b->d_major;
    k_majorant<<<(nbricks + 3) / 4, 256, 0, b->stream>>>(b->d_bricks, d_major, nbricks);
    hipError_t e = hipGetLastError();
    std::vector<float> h(nbricks);
    if (e == hipSuccess) e = hipMemcpyAsync(h.data(), d_major, (size_t)nbricks * sizeof(float), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (e != hipSuccess) return fail(b, VR_ERR_HIP, "majorant: %s", hipGetErrorString(e));
    // padding voxels hold the background, which getExtrema also sees for inactive voxels inside the bbox
    float m = h[0];
    for (uint32_t i = 1; i < nbricks; ++i) if (h[i] > m) m = h[i];
    b->max_value = m;
    {   // levels 1..4: maxima over 2^3 ... 16^3 bricks (on the host: at most 2^21 bricks)
        std::vector<float> lv(major_total - nbricks);
        for (int l = 1; l < VR_MAJOR_LEVELS; ++l) {
            const int* nc = b->major_nc[l];
            float* out = lv.data() + (b->major_off[l] - nbricks);
            for (int i = 0; i < nc[0]; ++i)
                for (int j = 0; j < nc[1]; ++j)
                    for (int k = 0; k < nc[2]; ++k) {
                        float mx = -INFINITY;
                        for (int di = 0; di < (1 << l); ++di)
                            for (int dj = 0; dj < (1 << l); ++dj)
                                for (int dk = 0; dk < (1 << l); ++dk) {
                                    const int bi = (i << l) + di, bj = (j << l) + dj, bk = (k << l) + dk;
                                    if (bi < g.nb[0] && bj < g.nb[1] && bk < g.nb[2]) mx = std::max(mx, h[((size_t)bi * g.nb[1] + bj) * g.nb[2] + bk]);
                                }
                        out[((size_t)i * nc[1] + j) * nc[2] + k] = mx;
                    }
        }
        if (!lv.empty()) HIP_TRY(b, hipMemcpy(b->d_major + nbricks, lv.data(), lv.size() * sizeof(float), hipMemcpyHostToDevice));
    }

    if (b->d_skip) { (void)hipFree(b->d_skip); b->d_skip = nullptr; }
    uint8_t *d_a = nullptr, *d_b = nullptr;
    e = hipMalloc((void**)&b->d_skip, total * sizeof(float));
    if (e == hipSuccess) e = hipMalloc((void**)&d_a, total);
    if (e == hipSuccess) e = hipMalloc((void**)&d_b, total);
    if (e == hipSuccess) {
        const int blocks = grid_blocks(total, 65536);
        k_skip_pass<<<blocks, 256, 0, b->stream>>>(b->d_bricks, nullptr, d_a, g.dim[0], g.dim[1], g.dim[2], g.nb[0], g.nb[1], g.nb[2], 2, VR_SKIP_CAP);
        k_skip_pass<<<blocks, 256, 0, b->stream>>>(b->d_bricks, d_a, d_b, g.dim[0], g.dim[1], g.dim[2], g.nb[0], g.nb[1], g.nb[2], 1, VR_SKIP_CAP);
        k_skip_pass<<<blocks, 256, 0, b->stream>>>(b->d_bricks, d_b, d_a, g.dim[0], g.dim[1], g.dim[2], g.nb[0], g.nb[1], g.nb[2], 0, VR_SKIP_CAP);
        k_skip_make<<<blocks, 256, 0, b->stream>>>(b->d_bricks, d_a, b->d_skip, total);
        e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    if (d_a) (void)hipFree(d_a);
    if (d_b) (void)hipFree(d_b);
    if (e != hipSuccess) return fail(b, e == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP, "skip grid: %s", hipGetErrorString(e));
    return VR_OK;
}

void set_map(vr_backend* b, const float* inv_mat, const float* vec, const float* mat, const int32_t* bmin, const int32_t* bmax)
{
    memcpy(b->inv_mat, inv_mat, sizeof b->inv_mat);
    memcpy(b->mat, mat, sizeof b->mat);
    memcpy(b->vec, vec, sizeof b->vec);
    for (int i = 0; i < 3; ++i) { b->bmin[i] = (float)bmin[i]; b->bmax1[i] = (float)(bmax[i] + 1); }   // Ray.h:401
}

// ---- threaded BVH (replaces the Embree scene, SceneLoader.cpp:150-199) ----
struct BuildTri { float lo[3], hi[3], c[3]; uint32_t prim; };

// The tree is built once (median split on the longest centroid axis, one triangle per leaf: a triangle test costs twice a box test, and leaves of 4 measured 5 % slower) and laid out
// eight times, once per sign octant of the ray direction: a threaded BVH visits children in array order, so
// each octant gets the order in which ITS rays meet the nearer child first and the far one can be culled
// against the hit already found.
struct TreeNode { BvhNode box; int axis; uint32_t left, right; };

// Where to split [first, first + count): the surface-area heuristic over every position along every axis (the triangles sorted by
// centroid; cost = area(left box) * n_left + area(right box) * n_right), or — `median`, and wherever the heuristic has nothing to
// choose from — the median along the longest centroid axis.  Leaves `t` sorted along the returned axis.
uint32_t choose_split(std::vector<BuildTri>& t, uint32_t first, uint32_t count, int longest, bool median, int& axis_out)
{
    auto by_axis = [](int axis) { return [axis](const BuildTri& x, const BuildTri& y) { return x.c[axis] < y.c[axis] || (x.c[axis] == y.c[axis] && x.prim < y.prim); }; };
    axis_out = longest;
    if (median || count < 4) {
        const uint32_t mid = count / 2;
        std::nth_element(t.begin() + first, t.begin() + first + mid, t.begin() + first + count, by_axis(longest));
        return mid;
    }
    auto area = [](const float* lo, const float* hi) {
        const double dx = (double)hi[0] - lo[0], dy = (double)hi[1] - lo[1], dz = (double)hi[2] - lo[2];
        return dx * dy + dy * dz + dz * dx;
    };
    double   best = INFINITY;
    uint32_t best_k = count / 2;
    int      best_axis = longest;
    std::vector<double> right(count);
    for (int axis = 0; axis < 3; ++axis) {
        std::sort(t.begin() + first, t.begin() + first + count, by_axis(axis));
        float lo[3] = { INFINITY, INFINITY, INFINITY }, hi[3] = { -INFINITY, -INFINITY, -INFINITY };
        for (uint32_t i = count; i-- > 1;) {          // right[i] = area of the box of triangles i .. count - 1
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], t[first + i].lo[a]); hi[a] = std::max(hi[a], t[first + i].hi[a]); }
            right[i] = area(lo, hi);
        }
        for (int a = 0; a < 3; ++a) { lo[a] = INFINITY; hi[a] = -INFINITY; }
        for (uint32_t k = 1; k < count; ++k) {        // left = triangles 0 .. k - 1
            for (int a = 0; a < 3; ++a) { lo[a] = std::min(lo[a], t[first + k - 1].lo[a]); hi[a] = std::max(hi[a], t[first + k - 1].hi[a]); }
            const double cost = area(lo, hi) * k + right[k] * (count - k);
            if (cost < best) { best = cost; best_k = k; best_axis = axis; }
        }
    }
    if (best_axis != 2) std::sort(t.begin() + first, t.begin() + first + count, by_axis(best_axis));
    axis_out = best_axis;
    return best_k;
}

uint32_t build_bvh(std::vector<BuildTri>& t, uint32_t first, uint32_t count, std::vector<TreeNode>& tree, std::vector<uint32_t>& order,
                   uint32_t leaf_size, bool median)
{
    TreeNode tn{};
    BvhNode& nd = tn.box;
    float    clo[3] = { INFINITY, INFINITY, INFINITY }, chi[3] = { -INFINITY, -INFINITY, -INFINITY };
    for (int a = 0; a < 3; ++a) { nd.lo[a] = INFINITY; nd.hi[a] = -INFINITY; }
    for (uint32_t i = first; i < first + count; ++i)
        for (int a = 0; a < 3; ++a) {
            nd.lo[a] = std::min(nd.lo[a], t[i].lo[a]); nd.hi[a] = std::max(nd.hi[a], t[i].hi[a]);
            clo[a] = std::min(clo[a], t[i].c[a]); chi[a] = std::max(chi[a], t[i].c[a]);
        }
    // conservative pad: Moeller-Trumbore accepts points a few ulp outside the triangle's box
    for (int a = 0; a < 3; ++a) {
        const float ext = nd.hi[a] - nd.lo[a];
        const float mag = std::max(std::fabs(nd.lo[a]), std::fabs(nd.hi[a]));
        const float pad = 1e-4f * std::max(ext, 1e-3f) + 1e-5f * mag;
        nd.lo[a] -= pad; nd.hi[a] += pad;
    }
    const uint32_t me = (uint32_t)tree.size();
    tree.push_back(tn);
    int axis = 0;
    for (int a = 1; a < 3; ++a) if (chi[a] - clo[a] > chi[axis] - clo[axis]) axis = a;
    if (count <= leaf_size || (!(chi[axis] > clo[axis]) && count <= 255u)) {   // (coincident centroids: one leaf, of at most 255 — the node's count has 8 bits)
        tree[me].box.leaf = (count << 24) | (uint32_t)order.size();   // (set_mesh checks: < 2^24 triangles, leaves of < 256)
        for (uint32_t i = first; i < first + count; ++i) order.push_back(i);
    } else {
        int            split_axis = axis;
        const uint32_t mid = choose_split(t, first, count, axis, median, split_axis);
        tree[me].box.leaf = 0;
        tree[me].axis = split_axis;
        const uint32_t l = build_bvh(t, first, mid, tree, order, leaf_size, median);             // lower centroids along the split axis
        const uint32_t r = build_bvh(t, first + mid, count - mid, tree, order, leaf_size, median);
        tree[me].left = l; tree[me].right = r;
    }
    return me;
}

// depth-first layout for the rays of one sign octant (bit a set: direction component a is negative)
void thread_bvh(const std::vector<TreeNode>& tree, uint32_t n, int octant, std::vector<BvhNode>& nodes, size_t base)
{
    const size_t me = nodes.size();
    nodes.push_back(tree[n].box);
    if (tree[n].box.leaf == 0) {
        const bool neg = (octant >> tree[n].axis) & 1;
        thread_bvh(tree, neg ? tree[n].right : tree[n].left, octant, nodes, base);
        thread_bvh(tree, neg ? tree[n].left : tree[n].right, octant, nodes, base);
    }
    nodes[me].skip = (uint32_t)(nodes.size() - base);
}

struct Frame {
    CameraDev cam;
    uint32_t  spp;
};

// renderPartial's constants (BaseIntegrator.cpp:288-312) and Utils::rotate's matrix (Utils.cpp:30-53)
void make_camera(const vr_options_pod* o, CameraDev& c)
{
    memset(&c, 0, sizeof c);
    const float width = o->widthReference > 0.0f ? (float)o->widthReference : (float)o->width;
    const float height = o->heightReference > 0.0f ? (float)o->heightReference : (float)o->height;
    const float half_fov = (float)(o->fov * 0.5);
    const float rad = (float)(half_fov * M_PI / 180);       // Utils::deg2rad
    const float scale = tanf(rad);
    const float aspect = width / height;
    const float x = (float)((2 * (0.5) / width - 1) * aspect * scale);
    const float y = (float)((1 - 2 * (0.5) / height) * scale);
    const float x1 = (float)((2 * (1.5) / width - 1) * aspect * scale);
    const float y1 = (float)((1 - 2 * (1.5) / height) * scale);
    c.scale = scale; c.aspect = aspect; c.pixel_w = x1 - x; c.pixel_h = y1 - y;
    c.ref_w = o->widthReference > 0 ? o->widthReference : o->width;
    c.ref_h = o->heightReference > 0 ? o->heightReference : o->height;
    c.x0 = o->widthStartOffset; c.y0 = o->heightStartOffset;
    c.width = o->width;
    for (int i = 0; i < 3; ++i) c.pos[i] = o->cameraPosition[i];
    c.has_rot = (o->cameraRotation[1] != 0 || o->cameraRotation[0] != 0 || o->cameraRotation[2] != 0) ? 1 : 0;  // BaseIntegrator.cpp:56
    if (c.has_rot) {
        // rotate(pitch = rot.y, roll = rot.x, yaw = rot.z)                                       BaseIntegrator.cpp:57
        float pitch = o->cameraRotation[1], roll = o->cameraRotation[0], yaw = o->cameraRotation[2];
        pitch = (float)((pitch * M_PI) / 180); roll = (float)((roll * M_PI) / 180); yaw = (float)((yaw * M_PI) / 180);
        const float cosa = (float)cos((double)yaw), sina = (float)sin((double)yaw);
        const float cosb = (float)cos((double)pitch), sinb = (float)sin((double)pitch);
        const float cosc = (float)cos((double)roll), sinc = (float)sin((double)roll);
        c.rot[0] = cosa * cosb; c.rot[1] = cosa * sinb * sinc - sina * cosc; c.rot[2] = cosa * sinb * cosc + sina * sinc;
        c.rot[3] = sina * cosb; c.rot[4] = sina * sinb * sinc + cosa * cosc; c.rot[5] = sina * sinb * cosc - cosa * sinc;
        c.rot[6] = -sinb;       c.rot[7] = cosb * sinc;                      c.rot[8] = cosb * cosc;
    }
    c.spp = o->rayPerPixelCount;   // c.offsets: k_quadrant_offsets fills them on the device (render_row_set)
}

// developer knob: positive integer from the environment, else 0 (= the built-in default)
int tune_env(const char* name)
{
    const char* v = getenv(name);
    if (!v || !*v) return 0;
    const int x = atoi(v);
    return x > 0 ? x : 0;
}
inline int tuned(int knob, int dflt) { return knob > 0 ? knob : dflt; }
// the VR_TUNE_* knobs, read once per render call / mesh upload (not per pass or per BVH node)
void read_knobs(vr_backend* b)
{
    auto plus1 = [](const char* name) { const char* v = getenv(name); return v && *v ? atoi(v) + 1 : 0; };   // 0 is a valid value
    b->tune_tile_h_log = plus1("VR_TUNE_TILE_H_LOG");
    b->tune_march_tile_h_log = plus1("VR_TUNE_MARCH_TILE_H_LOG");
    b->tune_major_level = plus1("VR_TUNE_MAJORANT_LEVEL");
    b->tune_round_max = tune_env("VR_TUNE_ROUND_MAX"); b->tune_round_events = tune_env("VR_TUNE_ROUND_EVENTS");
    b->tune_chunk = tune_env("VR_TUNE_CHUNK"); b->tune_bvh_leaf = tune_env("VR_TUNE_BVH_LEAF"); b->tune_bvh_median = tune_env("VR_TUNE_BVH_MEDIAN");
    b->tune_no_geom = tune_env("VR_TUNE_NO_GEOM"); b->tune_no_lds_major = tune_env("VR_TUNE_NO_LDS_MAJOR");
}

float ulp_of(float v)
{
    v = std::fabs(v);
    return std::nextafter(v, INFINITY) - v;
}

// upper estimate of the ray parameter t (index units) any walk of this frame can reach
float tracking_t_max(const vr_backend* b, const vr_options_pod* o)
{
    float diag = 0.0f, dist = 0.0f;
    for (int i = 0; i < 3; ++i) {
        const float ext = b->bmax1[i] - b->bmin[i];
        diag += ext * ext;
        float e = 0.0f;
        for (int k = 0; k < 3; ++k) e += b->inv_mat[3 * i + k] * (o->cameraPosition[k] - b->vec[k]);
        const float c = 0.5f * (b->bmax1[i] + b->bmin[i]);
        dist += (e - c) * (e - c);
    }
    return std::sqrt(dist) + 2.0f * std::sqrt(diag);
}

int validate(vr_backend* b, const vr_options_pod* o, int integrator)
{
    if (!o) return fail(b, VR_ERR_INVALID_ARG, "options == NULL");
    if (integrator < 0 || integrator >= VR_INTEGRATOR_COUNT) return fail(b, VR_ERR_INVALID_ARG, "unknown integrator id %d", integrator);
    if (o->width == 0 || o->height == 0 || o->rayPerPixelCount == 0) return fail(b, VR_ERR_INVALID_ARG, "width, height and rayPerPixelCount must be > 0");
    if (o->rayPerPixelCount > (1u << 20)) return fail(b, VR_ERR_INVALID_ARG, "rayPerPixelCount above 2^20 (the sample index has 20 bits in the kernels' stream id)");
    if (o->width + o->widthStartOffset > 65536u || o->height + o->heightStartOffset > 65536u)
        return fail(b, VR_ERR_INVALID_ARG, "image coordinates must stay below 65536 (Philox pixel key)");
    if (o->maxDepth > 255) return fail(b, VR_ERR_INVALID_ARG, "maxDepth is a uint8_t in the reference (Types.h:101)");
    if (integrator == VR_INTEGRATOR_NANOVDB_SIMPLE || integrator == VR_INTEGRATOR_NANOVDB_EMISSION) {
        if (!b->has_density) return fail(b, VR_ERR_NO_DENSITY, "the grid marchers need a density grid (SceneLoader.cpp:212-213)");
        if (integrator == VR_INTEGRATOR_NANOVDB_EMISSION && !b->has_temperature)
            return fail(b, VR_ERR_NO_DENSITY, "nanoVDBEmission needs a temperature grid (IntegratorNanoVDBEmission.cpp:13-18 exits)");
    } else if (integrator == VR_INTEGRATOR_RATIO_TRACKING || integrator == VR_INTEGRATOR_DELTA_TRACKING ||
               integrator == VR_INTEGRATOR_MESH_RATIO_TRACKING || integrator == VR_INTEGRATOR_MESH_DELTA_TRACKING) {
        if (!b->has_density) return fail(b, VR_ERR_NO_DENSITY, "tracking integrators need a density grid (SceneLoader.cpp:212-213)");
        if ((integrator == VR_INTEGRATOR_MESH_RATIO_TRACKING || integrator == VR_INTEGRATOR_MESH_DELTA_TRACKING) && !b->has_mesh)
            return fail(b, VR_ERR_NO_MESH, "the mesh-bounded trackers need a mesh");
        // every step must advance t in float, or the reference's loops never end
        if (!(o->stepSizeMin > 0.0f) || !(o->stepSizeMax >= o->stepSizeMin))
            return fail(b, VR_ERR_INVALID_ARG, "stepSize min must be > 0 and <= max (the reference would not terminate)");
        const float tmax = tracking_t_max(b, o);
        float diag = 0.0f;
        for (int i = 0; i < 3; ++i) diag += (b->bmax1[i] - b->bmin[i]) * (b->bmax1[i] - b->bmin[i]);
        if (o->stepSizeMin < ulp_of(tmax))
            return fail(b, VR_ERR_INVALID_ARG, "stepSize min %g is below float resolution at ray parameter %g", o->stepSizeMin, tmax);
        if (std::sqrt(diag) / o->stepSizeMin > 1.0e8f)
            return fail(b, VR_ERR_INVALID_ARG, "stepSize min %g needs more than 1e8 steps across the grid", o->stepSizeMin);
        if ((b->mode == VR_MODE_FAST_LOCAL || b->mode == VR_MODE_EXACT_LOCAL) &&
            (integrator == VR_INTEGRATOR_DELTA_TRACKING || integrator == VR_INTEGRATOR_MESH_DELTA_TRACKING)) {
            // per-brick majorants change the reference's estimator wherever it scales or clamps its free paths
            if (o->stepSizeMultiplier != 1.0f || o->stepSizeMin > 0.01f || o->stepSizeMax < 1.0e4f)
                return fail(b, VR_ERR_UNSUPPORTED, "VR_MODE_FAST_LOCAL needs the textbook regime (stepSizeMultiplier 1, stepSizeMin <= 0.01, "
                            "stepSizeMax >= 1e4): with scaled or clamped free paths the reference is not a delta tracker and local "
                            "majorants would change its image");
            if (o->useImportanceSampling) return fail(b, VR_ERR_UNSUPPORTED, "VR_MODE_FAST_LOCAL does not track the path pdf (useImportanceSampling)");
            if (!(o->lightRayDensityMultiplier >= 0.0f)) return fail(b, VR_ERR_UNSUPPORTED, "VR_MODE_FAST_LOCAL needs lightRayDensityMultiplier >= 0");
        }
    } else {
        if (!b->has_mesh) return fail(b, VR_ERR_NO_MESH, "marching integrators need a mesh");
    }
    return VR_OK;
}

template <typename K, typename P>
int launch_cap(vr_backend* b, K kernel, const P&, int& blocks_per_cu)
{
    int nb = 0;
    for (int i = 0; i < b->noccupancy; ++i)
        if (b->occupancy[i].kernel == (const void*)kernel) nb = b->occupancy[i].blocks;
    if (nb == 0) {
        HIP_TRY(b, hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, kernel, 256, 0));
        if (b->noccupancy < 64) { b->occupancy[b->noccupancy].kernel = (const void*)kernel; b->occupancy[b->noccupancy].blocks = nb; b->noccupancy++; }
    }
    blocks_per_cu = std::max(1, std::min(nb, 8));
    return VR_OK;
}

// noalloc: the allocation-free variant — every scratch buffer comes from this device's copy of `scratch`, nothing is
// allocated or freed, no event is created (so no kernel time is reported) and nothing is copied from host memory.
int render_row_set(vr_backend* b, const vr_options_pod* o, int integrator, uint32_t row_first, uint32_t row_stride,
                   uint32_t row_count, float* d_out, bool noalloc)
{
    int rc = use_device(b);
    if (rc) return rc;
    rc = validate(b, o, integrator);
    if (rc) return rc;
    if (row_count == 0) return VR_OK;
    if (row_stride == 0) return fail(b, VR_ERR_INVALID_ARG, "row_stride must be > 0");
    if ((uint64_t)row_first + (uint64_t)(row_count - 1) * row_stride >= o->height)
        return fail(b, VR_ERR_INVALID_ARG, "row set exceeds height %u", o->height);
    if (!d_out) return fail(b, VR_ERR_INVALID_ARG, "output == NULL");

    read_knobs(b);
    CameraDev cam;
    make_camera(o, cam);
    cam.row_first = row_first; cam.row_stride = row_stride; cam.nrows = row_count;

    const uint32_t spp = o->rayPerPixelCount;
    const uint64_t npix = (uint64_t)row_count * o->width;
    if (npix * 3 >= (1ull << 32)) return fail(b, VR_ERR_INVALID_ARG, "row set too large");
    const uint64_t budget = noalloc ? VR_NOALLOC_SAMPLE_BYTES : b->sample_budget;
    if (noalloc) {
        if (!b->d_noalloc) return fail(b, VR_ERR_HIP, "the scratch array of the allocation-free entry point was not resolved at create");
        if (npix * 12 > VR_NOALLOC_SAMPLE_BYTES)
            return fail(b, VR_ERR_INVALID_ARG, "row set of %llu pixels needs %llu bytes per sample pass; the allocation-free entry point "
                        "holds %llu (at most %llu pixels per call): render fewer rows per call",
                        (unsigned long long)npix, (unsigned long long)(npix * 12), (unsigned long long)VR_NOALLOC_SAMPLE_BYTES,
                        (unsigned long long)(VR_NOALLOC_SAMPLE_BYTES / 12));
        if (spp > VR_NOALLOC_MAX_SPP)
            return fail(b, VR_ERR_INVALID_ARG, "rayPerPixelCount %u above the %u the allocation-free entry point holds offsets for", spp, VR_NOALLOC_MAX_SPP);
    }
    uint64_t per_pass = std::max<uint64_t>(1, budget / (npix * 12));
    per_pass = std::min<uint64_t>(per_pass, spp);
    // The trackers hand out work tiles of 64 pixels, 8x8 over the row set, and their item space is padded to whole
    // tiles.  (On a rank's share of 8-way interleaved rows 8x8 still beat the flatter shapes that are squarer in the
    // image: 752 / 754 / 759 / 758 ms for 8x8 / 16x4 / 32x2 / 64x1 on rank 3's share of the N=8 headline frame.)
    const int tile_h_log = b->tune_tile_h_log > 0 ? b->tune_tile_h_log - 1 : 3;      // (the knob is stored + 1: 0 is a valid value)
    if (tile_h_log < 0 || tile_h_log > 6) return fail(b, VR_ERR_INVALID_ARG, "VR_TUNE_TILE_H_LOG must be in [0, 6]");
    const uint64_t tile_w = 64u >> tile_h_log, tile_h = 1u << tile_h_log;
    const uint64_t npix_tiled = ((o->width + tile_w - 1) / tile_w) * ((row_count + tile_h - 1) / tile_h) * 64u;
    // item indices are 32-bit and the work counters run past the last item by a chunk per wave and strip: half the range is spare
    per_pass = std::min<uint64_t>(per_pass, std::max<uint64_t>(1, 0x7fffffffull / npix_tiled));
    if (npix_tiled * per_pass >= (1ull << 31)) return fail(b, VR_ERR_INVALID_ARG, "row set too large for one pass");

    // scratch of this call
    float*     d_samples;
    float2*    d_offsets;
    uint32_t*  d_work;
    Counters*  d_counters;
    TrackCold* d_cold;
    if (noalloc) {
        d_work = (uint32_t*)(b->d_noalloc + VR_NOALLOC_OFF_WORK);
        d_counters = (Counters*)(b->d_noalloc + VR_NOALLOC_OFF_COUNTERS);
        d_cold = (TrackCold*)(b->d_noalloc + VR_NOALLOC_OFF_COLD);
        d_offsets = (float2*)(b->d_noalloc + VR_NOALLOC_OFF_OFFSETS);
        d_samples = (float*)(b->d_noalloc + VR_NOALLOC_HEADER_BYTES);
    } else {
        // the staging buffer grows to what the frame needs; when HBM does not have it (other users of the GPU, a
        // smaller device) render in more passes rather than fail
        for (;;) {
            rc = ensure(b, b->d_samples, b->samples_cap, (size_t)(npix * per_pass * 3));
            if (rc != VR_ERR_OUT_OF_MEMORY || per_pass == 1) break;
            (void)hipGetLastError();
            per_pass = (per_pass + 1) / 2;
        }
        if (rc) return rc;
        rc = ensure(b, b->d_offsets, b->offsets_cap, (size_t)spp);
        if (rc) return rc;
        d_work = b->d_work; d_counters = b->d_counters; d_cold = b->d_cold; d_offsets = b->d_offsets; d_samples = b->d_samples;
    }
    cam.offsets = d_offsets;
    k_quadrant_offsets<<<(spp + 255) / 256 > 256 ? 256 : (spp + 255) / 256, 256, 0, b->stream>>>(d_offsets, spp);
    HIP_TRY(b, hipGetLastError());

    // the voxel-span / empty-run estimates assume ray parameters of at most a few thousand voxels
    bool span_ok = true;
    const bool mesh_bound = integrator == VR_INTEGRATOR_MESH_RATIO_TRACKING || integrator == VR_INTEGRATOR_MESH_DELTA_TRACKING;
    const bool tracking = integrator == VR_INTEGRATOR_RATIO_TRACKING || integrator == VR_INTEGRATOR_DELTA_TRACKING || mesh_bound;
    if (tracking) span_ok = tracking_t_max(b, o) < 8000.0f;

    WorkDev work{};
    work.counter = d_work; work.samples = d_samples; work.counters = d_counters;
    work.key0 = (uint32_t)b->seed; work.key1 = (uint32_t)(b->seed >> 32);
    work.step_limit = 1u << 28;

    HIP_TRY(b, hipMemsetAsync(d_counters, 0, sizeof(Counters), b->stream));
    b->last_counters = d_counters;
    b->passes_used = 0;
    b->launches = 0;
    const bool exact = b->mode != VR_MODE_FAST && b->mode != VR_MODE_FAST_LOCAL;
    const bool local_majorants = (b->mode == VR_MODE_FAST_LOCAL || b->mode == VR_MODE_EXACT_LOCAL) &&
                                 (integrator == VR_INTEGRATOR_DELTA_TRACKING || integrator == VR_INTEGRATOR_MESH_DELTA_TRACKING);
    const bool timed = !noalloc;

    for (uint32_t s0 = 0; s0 < spp; s0 += (uint32_t)per_pass) {
        const uint32_t sc = (uint32_t)std::min<uint64_t>(per_pass, spp - s0);
        cam.s_first = s0; cam.s_count = sc;
        work.nitems = (uint32_t)(npix * sc);
        Pass ev{};
        if (timed) {
            if (b->passes_used == b->passes.size()) {
                Pass p{};
                HIP_TRY(b, hipEventCreate(&p.e0));
                HIP_TRY(b, hipEventCreate(&p.e1));
                b->passes.push_back(p);
            }
            ev = b->passes[b->passes_used++];
        }
        b->launches++;
        HIP_TRY(b, hipMemsetAsync(d_work, 0, 256, b->stream));   // the item counter (grid marchers: one per strip, 32 B apart)

        if (integrator == VR_INTEGRATOR_NANOVDB_SIMPLE || integrator == VR_INTEGRATOR_NANOVDB_EMISSION) {
            GridMarchParams p{};
            p.cam = cam; p.work = work;
            p.density = b->grid; p.temperature = b->tgrid;
            memcpy(p.inv_mat, b->inv_mat, sizeof p.inv_mat); memcpy(p.vec, b->vec, sizeof p.vec); memcpy(p.mat, b->mat, sizeof p.mat);
            memcpy(p.bmin, b->bmin, sizeof p.bmin); memcpy(p.bmax1, b->bmax1, sizeof p.bmax1);
            p.g = o->heyneyGreensteinG;
            for (int i = 0; i < 3; ++i) {
                p.light[i] = o->lightPosition[i]; p.light_color[i] = o->lightColor[i];
                p.emission[i] = o->emissionColor[i]; p.background[i] = o->backgroundColor[i];
            }
            void (*kern)(const GridMarchParams) =
                integrator == VR_INTEGRATOR_NANOVDB_SIMPLE
                    ? (exact ? vr_gridmarch_kernel<false, true> : vr_gridmarch_kernel<false, false>)
                    : (exact ? vr_gridmarch_kernel<true, true> : vr_gridmarch_kernel<true, false>);
            // persistent waves, each drawing 8x8-pixel tiles from its strip's counter
            int per_cu = 8;
            rc = launch_cap(b, kern, p, per_cu);
            if (rc) return rc;
            const uint64_t ntiles = (uint64_t)((cam.width + 7) / 8) * ((cam.nrows + 7) / 8) * sc;
            const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)per_cu * b->num_cu, (ntiles + 3) / 4));
            if (timed) HIP_TRY(b, hipEventRecord(ev.e0, b->stream));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, b->stream, p);
            HIP_TRY(b, hipGetLastError());
            if (timed) HIP_TRY(b, hipEventRecord(ev.e1, b->stream));
        } else if (tracking) {
            TrackCold cold{};
            cold.cam = cam;
            memcpy(cold.inv_mat, b->inv_mat, sizeof cold.inv_mat); memcpy(cold.vec, b->vec, sizeof cold.vec);
            memcpy(cold.bmin, b->bmin, sizeof cold.bmin); memcpy(cold.bmax1, b->bmax1, sizeof cold.bmax1);
            cold.g = o->heyneyGreensteinG;
            for (int i = 0; i < 3; ++i) { cold.light[i] = o->lightPosition[i]; cold.light_color[i] = o->lightColor[i]; }
            for (int i = 0; i < 3; ++i) {
                cold.emission[i] = o->emissionColor[i]; cold.medium[i] = o->mediumColor[i]; cold.background[i] = o->backgroundColor[i];
            }
            cold.max_depth = o->maxDepth;
            cold.tile_h_log = (uint32_t)tile_h_log;
            work.nitems = (uint32_t)(npix_tiled * sc);
            cold.temperature = b->tgrid;
            if (mesh_bound) {
                cold.mesh.nodes = b->d_nodes; cold.mesh.tris = b->d_tris; cold.mesh.prim_to_shape = b->d_p2s;
                cold.mesh.shape_dissolve = b->d_dissolve; cold.mesh.shape_diffuse = b->d_diffuse; cold.mesh.nnodes = b->nnodes;
                memcpy(cold.mat, b->mat, sizeof cold.mat);
            }
            k_store_cold<<<1, 64, 0, b->stream>>>(cold, d_cold);
            HIP_TRY(b, hipGetLastError());
            if (local_majorants) {
                DdaParams dp{};
                dp.cold = d_cold; dp.work = work;
                dp.density = b->grid;
                // majorant cells of 32^3 voxels (level 2) by default: measured 2128 / 2591 / 2725 / 2350 / 1627 M samples/s for levels 0-4 on
                // config 3 textbook and 1336 / 1720 / 1898 / 1812 / 1421 on config 4 (VR_TUNE_MAJORANT_LEVEL; 0 = the 8^3 bricks)
                // (a sixteenth of the grid's largest extent, between 8 and 32 voxels: small grids keep the bricks)
                const int maxdim = std::max(b->grid.dim[0], std::max(b->grid.dim[1], b->grid.dim[2]));
                // (from 1024 voxels on: 64^3-voxel cells, whose grid of at most 4096 cells the kernel keeps in LDS — measured on
                // config 4: 4405 M samples/s against 4351 for 32^3-voxel cells read through L2 and 4345 for 64^3 through L2)
                const int auto_level = maxdim >= 1024 ? 3 : (maxdim >= 512 ? 2 : (maxdim >= 256 ? 1 : 0));
                const int level = b->tune_major_level > 0 ? std::min(b->tune_major_level - 1, VR_MAJOR_LEVELS - 1) : auto_level;
                dp.major = b->d_major + b->major_off[level]; dp.cell_log = 3 + level;
                for (int a = 0; a < 3; ++a) dp.nc[a] = b->major_nc[level][a];
                dp.sigma_a = o->sigma_a; dp.sigma_s = o->sigma_s; dp.light_mult = o->lightRayDensityMultiplier;
                // a round ends when 44 of the 64 lanes wait for the event code: the phases (path set-up, Philox, direction sample) cost far more
                // than the DDA steps, so they run as full as they can (swept: 2001 / 2720 / 3218 / 3574 / 3785 / 3765 / 3631 M samples/s for
                // 8 / 16 / 24 / 32 / 44 / 52 / 58 on config 3 textbook; the global-majorant kernels peak at 24)
                dp.round_max = tuned(b->tune_round_max, 64); dp.round_events = tuned(b->tune_round_events, mesh_bound ? 58 : 44);   // (mesh-bounded: the event code holds BVH walks — 1589 / 1485 / 1439 / 1440 ms at 44 / 52 / 58 / 62 on config 5's delta leg)
                const bool lds_major = !exact && !b->tune_no_lds_major && (uint64_t)dp.nc[0] * dp.nc[1] * dp.nc[2] <= VR_DDA_LDS_CELLS;
#define VR_DDA_K(T, E, L) (mesh_bound ? vr_delta_dda_kernel<T, E, L, true> : vr_delta_dda_kernel<T, E, L, false>)
                void (*dk)(const DdaParams) = exact ? (b->has_temperature ? VR_DDA_K(true, true, false) : VR_DDA_K(false, true, false))
                                            : lds_major ? (b->has_temperature ? VR_DDA_K(true, false, true) : VR_DDA_K(false, false, true))
                                                        : (b->has_temperature ? VR_DDA_K(true, false, false) : VR_DDA_K(false, false, false));
#undef VR_DDA_K
                int per_cu = 8;
                rc = launch_cap(b, dk, dp, per_cu);
                if (rc) return rc;
                const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)per_cu * b->num_cu, (work.nitems + 255ull) / 256));
                const uint64_t fair = work.nitems / ((uint64_t)blocks * 4 * 8);
                dp.work.chunk = (uint32_t)std::min<uint64_t>((uint64_t)tuned(b->tune_chunk, 128), std::max<uint64_t>(64, fair & ~63ull));
                if (timed) HIP_TRY(b, hipEventRecord(ev.e0, b->stream));
                hipLaunchKernelGGL(dk, dim3(blocks), dim3(256), 0, b->stream, dp);
                HIP_TRY(b, hipGetLastError());
                if (timed) HIP_TRY(b, hipEventRecord(ev.e1, b->stream));
            } else {
            TrackParams p{};
            p.cold = d_cold; p.work = work;
            p.density = b->grid;
            const bool temp = b->has_temperature;
            const bool is = o->useImportanceSampling != 0;
            p.sigma_a = o->sigma_a; p.sigma_s = o->sigma_s;
            p.sigma_maj = b->max_value * (o->sigma_a + o->sigma_s);          // IntegratorRatioTracking.cpp:42-45
            p.t_min = o->stepSizeMin; p.t_max = o->stepSizeMax; p.step_mult = o->stepSizeMultiplier;
            p.light_mult = o->lightRayDensityMultiplier; p.shadow_mult = o->shadowRayDensityMultiplier;
            // typical free path (index units) well under a voxel: remember each voxel's t-interval
            bool span = false;
            {
                const float mean_free = p.sigma_maj > 0.0f ? o->stepSizeMultiplier / p.sigma_maj : 1e30f;
                const float typical = std::min(std::max(o->stepSizeMin, mean_free), o->stepSizeMax);
                span = typical < 0.25f && span_ok;
            }
            const bool ratio = integrator == VR_INTEGRATOR_RATIO_TRACKING || integrator == VR_INTEGRATOR_MESH_RATIO_TRACKING;
            // a round of the step loop ends after round_max trips or when round_events lanes wait for the
            // event phase (measured per regime: sub-voxel steps have long trips and rare events, voxel-sized
            // steps one iteration per trip and an event every few iterations)
            // The mesh-bounded kernels' event phases hold the closest-hit queries (a BVH traversal per new ray, 1 200-3 600
            // instructions for the few lanes that wait), so their rounds run as long and as full as they can: 48 waiting lanes
            // and up to 96 trips (swept on config 5's tracker leg: ratio 129 -> 181 M samples/s, delta 450 -> 828 shipped,
            // 519 -> 1 010 textbook against the limits of the plain kernels).
            p.round_max = tuned(b->tune_round_max, (span && !mesh_bound) ? VR_ROUND_MAX : 2 * VR_ROUND_MAX);
            p.round_events = tuned(b->tune_round_events, mesh_bound ? (ratio ? 4 * VR_ROUND_EVENTS : 56) : (span ? VR_ROUND_EVENTS : 2 * VR_ROUND_EVENTS));   // (mesh-bounded delta tracking, re-swept with 32-byte nodes: 2073 / 1866 / 1720 / 1664 ms at 32 / 40 / 48 / 56 on config 5's textbook leg; the ratio leg is flat from 48 on)   // (the ratio tracker took 16 while a trip drew from the generator twice per iteration; 12 since: +1.5 %)
            void (*kern)(const TrackParams);
            // empty-space runs read the skip grid, whose empty voxels only keep the sign of sigma for
            // multipliers >= 0 (the shadow multiplier is only read by the ratio tracker)
            const bool skip_ok = span_ok && o->lightRayDensityMultiplier >= 0.0f && (!ratio || o->shadowRayDensityMultiplier >= 0.0f);
            if (skip_ok && b->mode != VR_MODE_EXACT) p.density.bricks = b->d_skip;   // (FAST_LOCAL renders the other integrators as FAST)
#define VR_TK5(I, E, S, Q, T, N) (mesh_bound ? vr_track_kernel<I, E, S, Q, T, N, true> : vr_track_kernel<I, E, S, Q, T, N, false>)
#define VR_TK4(I, E, S, Q) (temp ? (span ? VR_TK5(I, E, S, Q, true, true) : VR_TK5(I, E, S, Q, true, false)) \
                                 : (span ? VR_TK5(I, E, S, Q, false, true) : VR_TK5(I, E, S, Q, false, false)))
#define VR_TRACK_KERNEL(E, S) (ratio ? (is ? VR_TK4(INTEG_RATIO, E, S, true) : VR_TK4(INTEG_RATIO, E, S, false)) \
                                     : (is ? VR_TK4(INTEG_DELTA, E, S, true) : VR_TK4(INTEG_DELTA, E, S, false)))
            // (only the delta trackers have a local-majorant kernel: the LOCAL modes render the ratio trackers as FAST / EXACT_SKIP)
            const int kmode = b->mode == VR_MODE_FAST_LOCAL ? VR_MODE_FAST : (b->mode == VR_MODE_EXACT_LOCAL ? VR_MODE_EXACT_SKIP : b->mode);
            switch (skip_ok ? kmode : (kmode == VR_MODE_FAST ? -1 : VR_MODE_EXACT)) {
            case -1:                 kern = VR_TRACK_KERNEL(false, false); break;
            case VR_MODE_EXACT:      kern = VR_TRACK_KERNEL(true, false); break;
            case VR_MODE_EXACT_SKIP: kern = VR_TRACK_KERNEL(true, true); break;
            default:                 kern = VR_TRACK_KERNEL(false, true); break;
            }
            // FAST delta tracking with EVERY free path clamped to stepSizeMin (-ln(1e-5) = 11.513 is the largest draw,
            // IntegratorDeltaTracking.cpp:63-67; true for bunny-cloud-deltaTracking-configRenderer.xml:13): a voxel's steps in one
            // closed-form run (GEOM, vr_kernels.h).  VR_TUNE_NO_GEOM=1 keeps the step-by-step kernel for an A/B.
            const bool geom = skip_ok && kmode == VR_MODE_FAST && !ratio && !is && span && p.sigma_maj > 0.0f &&
                              11.52f * o->stepSizeMultiplier / p.sigma_maj <= o->stepSizeMin && !b->tune_no_geom;
            b->last_geom = geom;
            if (geom) {
                kern = temp ? (mesh_bound ? vr_track_kernel<INTEG_DELTA, false, true, false, true, true, true, true>
                                          : vr_track_kernel<INTEG_DELTA, false, true, false, true, true, false, true>)
                            : (mesh_bound ? vr_track_kernel<INTEG_DELTA, false, true, false, false, true, true, true>
                                          : vr_track_kernel<INTEG_DELTA, false, true, false, false, true, false, true>);
                // one run per trip and an event every few trips, like the voxel-sized-step kernels
                p.round_max = tuned(b->tune_round_max, 2 * VR_ROUND_MAX);
                p.round_events = tuned(b->tune_round_events, mesh_bound ? 56 : 2 * VR_ROUND_EVENTS);   // (swept: config 3 1223 / 1424 / 1476 / 1448 / 1302 M samples/s at 8 / 16 / 24 / 32 / 44; config 5's delta leg 752 / 1117 / 1185 at 24 / 48 / 56)
            }
#undef VR_TK5
#undef VR_TK4
#undef VR_TRACK_KERNEL
            int per_cu = 4;
            rc = launch_cap(b, kern, p, per_cu);
            if (rc) return rc;
            const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)per_cu * b->num_cu, (work.nitems + 255ull) / 256));
            // a wave's chunk: large enough that the counter's atomics do not serialise the launch, small
            // enough that the last chunks balance and that all waves stay on the same few image rows
            // (measured: 64-128 best on every workload, 2048 costs 8 % in L2 locality)
            const uint64_t fair = work.nitems / ((uint64_t)blocks * 4 * 8);
            p.work.chunk = (uint32_t)std::min<uint64_t>((uint64_t)tuned(b->tune_chunk, 128), std::max<uint64_t>(64, fair & ~63ull));
            if (timed) HIP_TRY(b, hipEventRecord(ev.e0, b->stream));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, b->stream, p);
            HIP_TRY(b, hipGetLastError());
            if (timed) HIP_TRY(b, hipEventRecord(ev.e1, b->stream));
            }
        } else {
            MarchParams p{};
            p.cam = cam; p.work = work;
            p.mesh.nodes = b->d_nodes; p.mesh.tris = b->d_tris; p.mesh.prim_to_shape = b->d_p2s;
            p.mesh.shape_dissolve = b->d_dissolve; p.mesh.shape_diffuse = b->d_diffuse; p.mesh.nnodes = b->nnodes;
            p.sigma_a = o->sigma_a; p.sigma_s = o->sigma_s; p.g = o->heyneyGreensteinG;
            for (int i = 0; i < 3; ++i) {
                p.light[i] = o->lightPosition[i]; p.light_color[i] = o->lightColor[i]; p.background[i] = o->backgroundColor[i];
            }
            p.max_depth = o->maxDepth;
            void (*kern)(const MarchParams) =
                integrator == VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_NEE
                    ? (exact ? vr_march_kernel<MARCH_NEE, true> : vr_march_kernel<MARCH_NEE, false>)
                : integrator == VR_INTEGRATOR_HOMOGENEOUS_RAY_MARCHER_IMPROVED
                    ? (exact ? vr_march_kernel<MARCH_IMPROVED, true> : vr_march_kernel<MARCH_IMPROVED, false>)
                : integrator == VR_INTEGRATOR_HETEROGENEOUS_PERLIN_NOISE
                    ? (exact ? vr_march_kernel<MARCH_PERLIN, true> : vr_march_kernel<MARCH_PERLIN, false>)
                    : (exact ? vr_march_kernel<MARCH_DENSITY, true> : vr_march_kernel<MARCH_DENSITY, false>);
            // grid-stride waves, each taking 8x8-pixel tiles: the item space is padded to whole tiles
            int per_cu = 8;
            rc = launch_cap(b, kern, p, per_cu);
            if (rc) return rc;
            // 8x8 over consecutive rows; 16x4 over a rank's share of interleaved rows, which is more compact in the
            // image (config 5, rank 3 of 8: 17.3 / 16.2 ms against 17.8 / 16.6 ms for 8x8; whole frame: 8x8 by 2 %)
            p.tile_h_log = (uint32_t)(b->tune_march_tile_h_log > 0 ? b->tune_march_tile_h_log - 1 : (cam.row_stride >= 4 ? 2 : 3));
            if (p.tile_h_log > 6) return fail(b, VR_ERR_INVALID_ARG, "VR_TUNE_MARCH_TILE_H_LOG must be in [0, 6]");
            const uint64_t mtw = 64u >> p.tile_h_log, mth = 1u << p.tile_h_log;
            const uint64_t ntiles = ((cam.width + mtw - 1) / mtw) * ((cam.nrows + mth - 1) / mth) * sc;
            if (ntiles * 64u >= (1ull << 32)) return fail(b, VR_ERR_INVALID_ARG, "row set too large for one pass");
            p.work.nitems = (uint32_t)(ntiles * 64u);
            const int blocks = (int)std::max<uint64_t>(1, std::min<uint64_t>((uint64_t)per_cu * b->num_cu, (ntiles + 3) / 4));
            if (timed) HIP_TRY(b, hipEventRecord(ev.e0, b->stream));
            hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, b->stream, p);
            HIP_TRY(b, hipGetLastError());
            if (timed) HIP_TRY(b, hipEventRecord(ev.e1, b->stream));
        }
        const uint32_t nchan = (uint32_t)(npix * 3);
        vr_resolve_kernel<<<(nchan + 255) / 256, 256, 0, b->stream>>>(d_samples, d_out, (uint32_t)npix, sc, s0 == 0 ? 1 : 0,
                                                                      s0 + sc >= spp ? 1 : 0, (float)spp);
        HIP_TRY(b, hipGetLastError());
    }
    return VR_OK;
}

int collect(vr_backend* b)
{
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    Counters c{};
    HIP_TRY(b, hipMemcpy(&c, b->last_counters ? b->last_counters : b->d_counters, sizeof c, hipMemcpyDeviceToHost));
    b->stats.path_samples = c.path_samples; b->stats.density_lookups = c.density_lookups;
    b->stats.scatter_events = c.scatter_events; b->stats.shadow_rays = c.shadow_rays;
    b->stats.ray_queries = c.ray_queries; b->stats.step_limit_hits = c.step_limit_hits;
    b->stats.wave_trips = c.wave_trips; b->stats.wave_trip_lanes = c.wave_trip_lanes; b->stats.event_phases = c.event_phases;
    b->stats.voxel_loads = c.voxel_loads;
    if (c.step_limit_hits) return fail(b, VR_ERR_STEP_LIMIT, "%llu paths hit the per-path step cap", (unsigned long long)c.step_limit_hits);
    return VR_OK;
}

} // namespace

// =============================================================================================
// C-ABI
// =============================================================================================
extern "C" {

void vr_abi_sizes(uint32_t out[4])
{
    out[0] = (uint32_t)sizeof(vr_options_pod); out[1] = (uint32_t)sizeof(vr_grid_desc);
    out[2] = (uint32_t)sizeof(vr_mesh_desc);   out[3] = (uint32_t)sizeof(vr_render_stats);
}

const char* vr_backend_last_error(const vr_backend* b) { return b ? b->err.c_str() : g_create_error.c_str(); }

int vr_backend_create(int device, vr_backend** out)
{
    if (!out) return fail(nullptr, VR_ERR_INVALID_ARG, "out == NULL");
    *out = nullptr;
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(nullptr, VR_ERR_NO_DEVICE, "no HIP device: %s", hipGetErrorString(e));
    if (device < 0 || device >= n) return fail(nullptr, VR_ERR_INVALID_ARG, "device %d out of range (0..%d)", device, n - 1);
    vr_backend* b = new vr_backend;
    b->device = device;
    auto bail = [&](const char* what, hipError_t err) {
        int code = fail(nullptr, err == hipErrorOutOfMemory ? VR_ERR_OUT_OF_MEMORY : VR_ERR_HIP, "%s: %s", what, hipGetErrorString(err));
        vr_backend_destroy(b);   // frees the stream and whatever was allocated so far
        return code;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail("hipSetDevice", e);
    hipDeviceProp_t prop;
    if ((e = hipGetDeviceProperties(&prop, device)) != hipSuccess) return bail("hipGetDeviceProperties", e);
    b->num_cu = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    if ((e = hipStreamCreateWithFlags(&b->stream, hipStreamNonBlocking)) != hipSuccess) return bail("hipStreamCreate", e);
    if ((e = hipMalloc((void**)&b->d_work, 256)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&b->d_counters, sizeof(Counters))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&b->d_maxbits, 256)) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMalloc((void**)&b->d_cold, sizeof(TrackCold))) != hipSuccess) return bail("hipMalloc", e);
    if ((e = hipMemset(b->d_counters, 0, sizeof(Counters))) != hipSuccess) return bail("hipMemset", e);
    // this device's copy of the allocation-free entry point's scratch array (each GPU holds its own, at its own address)
    if ((e = hipGetSymbolAddress((void**)&b->d_noalloc, HIP_SYMBOL(scratch))) != hipSuccess) return bail("hipGetSymbolAddress(scratch)", e);
    *out = b;
    return VR_OK;
}

void vr_backend_destroy(vr_backend* b)
{
    if (!b) return;
    (void)hipSetDevice(b->device);
    if (b->stream) (void)hipStreamSynchronize(b->stream);
    void* ptrs[] = { b->d_bricks, b->d_skip, b->d_major, b->d_tbricks, b->d_nodes, b->d_tris, b->d_p2s, b->d_dissolve, b->d_diffuse, b->d_samples,
                     b->d_rows, b->d_offsets, b->d_work, b->d_counters, b->d_maxbits, b->d_cold };
    for (void* p : ptrs) if (p) (void)hipFree(p);
    for (Pass& p : b->passes) { (void)hipEventDestroy(p.e0); (void)hipEventDestroy(p.e1); }
    if (b->stream) (void)hipStreamDestroy(b->stream);
    delete b;
}

int vr_backend_set_mode(vr_backend* b, int mode)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (mode != VR_MODE_EXACT && mode != VR_MODE_FAST && mode != VR_MODE_EXACT_SKIP && mode != VR_MODE_FAST_LOCAL && mode != VR_MODE_EXACT_LOCAL)
        return fail(b, VR_ERR_INVALID_ARG, "unknown mode %d", mode);
    b->mode = mode;
    return VR_OK;
}
int vr_backend_set_seed(vr_backend* b, uint64_t seed)
{
    if (!b) return VR_ERR_INVALID_ARG;
    b->seed = seed;
    return VR_OK;
}
int vr_backend_set_sample_buffer_bytes(vr_backend* b, uint64_t bytes)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (bytes < 12) return fail(b, VR_ERR_INVALID_ARG, "sample buffer must hold at least one sample");
    b->sample_budget = bytes;
    return VR_OK;
}

int vr_backend_set_density(vr_backend* b, const vr_grid_desc* density, const vr_grid_desc* temperature)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (!density) return fail(b, VR_ERR_INVALID_ARG, "density == NULL");
    int rc = use_device(b);
    if (rc) return rc;
    b->has_density = false; b->has_temperature = false;
    rc = upload_grid(b, density, b->d_bricks, b->grid);
    if (rc) return rc;
    set_map(b, density->inv_mat, density->vec, density->mat, density->bbox_min, density->bbox_max);
    rc = build_derived(b);
    if (rc) return rc;
    if (density->max_value >= 0.0f) b->max_value = density->max_value;   // the caller's getExtrema result
    if (temperature) {
        rc = upload_grid(b, temperature, b->d_tbricks, b->tgrid);
        if (rc) return rc;
        b->has_temperature = true;
    }
    b->has_density = true;
    return VR_OK;
}

int vr_backend_generate_cloud(vr_backend* b, int n, float radius_voxels)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (n < 8 || n > 1536 || (n & 7)) return fail(b, VR_ERR_INVALID_ARG, "n must be a multiple of 8 in [8, 1536]");
    if (!(radius_voxels > 0.0f)) return fail(b, VR_ERR_INVALID_ARG, "radius must be > 0");
    int rc = use_device(b);
    if (rc) return rc;
    b->has_density = false; b->has_temperature = false;
    const int    nb = n / 8;
    const size_t nvox = (size_t)n * n * n;
    if (b->d_bricks) { (void)hipFree(b->d_bricks); b->d_bricks = nullptr; }
    HIP_TRY(b, hipMalloc((void**)&b->d_bricks, nvox * sizeof(float)));
    k_cloud<<<grid_blocks(nvox, 16384), 256, 0, b->stream>>>(b->d_bricks, n, nb, radius_voxels);
    HIP_TRY(b, hipGetLastError());
    b->grid.bricks = b->d_bricks;
    int32_t bmin[3], bmax[3];
    for (int i = 0; i < 3; ++i) {
        b->grid.min[i] = -(n / 2); b->grid.dim[i] = n; b->grid.nb[i] = nb;
        bmin[i] = -(n / 2); bmax[i] = -(n / 2) + n - 1;
    }
    b->grid.background = 0.0f;
    const float ident[9] = { 1, 0, 0, 0, 1, 0, 0, 0, 1 }, zero[3] = { 0, 0, 0 };
    set_map(b, ident, zero, ident, bmin, bmax);
    rc = build_derived(b);
    if (rc) return rc;
    b->has_density = true;
    return VR_OK;
}

int vr_backend_set_density_map(vr_backend* b, float voxel_size, const float origin_xyz[3])
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (!b->has_density) return fail(b, VR_ERR_NO_DENSITY, "no density grid");
    if (!(voxel_size > 0.0f) || !origin_xyz) return fail(b, VR_ERR_INVALID_ARG, "voxel_size must be > 0 and origin not NULL");
    const float inv = 1.0f / voxel_size;
    for (int i = 0; i < 9; ++i) { b->mat[i] = (i % 4 == 0) ? voxel_size : 0.0f; b->inv_mat[i] = (i % 4 == 0) ? inv : 0.0f; }
    for (int i = 0; i < 3; ++i) b->vec[i] = origin_xyz[i];
    return VR_OK;
}

int vr_backend_density_max(vr_backend* b, float* max_value)
{
    if (!b || !max_value) return VR_ERR_INVALID_ARG;
    if (!b->has_density) return fail(b, VR_ERR_NO_DENSITY, "no density grid");
    *max_value = b->max_value;
    return VR_OK;
}

int vr_backend_read_density(vr_backend* b, float* voxels, uint64_t capacity, int32_t dims[3], int32_t bbox_min[3])
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (!b->has_density) return fail(b, VR_ERR_NO_DENSITY, "no density grid");
    int rc = use_device(b);
    if (rc) return rc;
    const GridDev& g = b->grid;
    if (dims) for (int i = 0; i < 3; ++i) dims[i] = g.dim[i];
    if (bbox_min) for (int i = 0; i < 3; ++i) bbox_min[i] = g.min[i];
    if (!voxels) return VR_OK;
    const size_t nvox = (size_t)g.dim[0] * g.dim[1] * g.dim[2];
    if (capacity < nvox) return fail(b, VR_ERR_INVALID_ARG, "buffer holds %llu floats, grid has %llu", (unsigned long long)capacity, (unsigned long long)nvox);
    float* d_dense = nullptr;
    HIP_TRY(b, hipMalloc((void**)&d_dense, nvox * sizeof(float)));
    k_unbrick<<<grid_blocks(nvox), 256, 0, b->stream>>>(g.bricks, d_dense, g.dim[0], g.dim[1], g.dim[2], g.nb[1], g.nb[2]);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(voxels, d_dense, nvox * sizeof(float), hipMemcpyDeviceToHost, b->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(b->stream);
    (void)hipFree(d_dense);
    if (e != hipSuccess) return fail(b, VR_ERR_HIP, "read_density: %s", hipGetErrorString(e));
    return VR_OK;
}

int vr_backend_set_mesh(vr_backend* b, const vr_mesh_desc* m)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (!m || !m->verts_xyz || !m->tri_idx || !m->prim_to_shape || !m->shape_dissolve || m->ntris == 0 || m->nshapes == 0)
        return fail(b, VR_ERR_INVALID_ARG, "incomplete mesh description");
    int rc = use_device(b);
    if (rc) return rc;
    b->has_mesh = false;
    std::vector<BuildTri> bt(m->ntris);
    for (uint32_t t = 0; t < m->ntris; ++t) {
        if (m->prim_to_shape[t] >= m->nshapes) return fail(b, VR_ERR_INVALID_ARG, "prim_to_shape[%u] out of range", t);
        for (int a = 0; a < 3; ++a) { bt[t].lo[a] = INFINITY; bt[t].hi[a] = -INFINITY; bt[t].c[a] = 0.0f; }
        for (int v = 0; v < 3; ++v) {
            const uint32_t vi = m->tri_idx[3 * (size_t)t + v];
            if (vi >= m->nverts) return fail(b, VR_ERR_INVALID_ARG, "triangle %u references vertex %u of %u", t, vi, m->nverts);
            for (int a = 0; a < 3; ++a) {
                const float x = m->verts_xyz[3 * (size_t)vi + a];
                bt[t].lo[a] = std::min(bt[t].lo[a], x); bt[t].hi[a] = std::max(bt[t].hi[a], x);
                bt[t].c[a] += x * (1.0f / 3.0f);
            }
        }
        bt[t].prim = t;
    }
    std::vector<TreeNode> tree;
    std::vector<uint32_t> order;
    read_knobs(b);
    if (m->ntris >= (1u << 24)) return fail(b, VR_ERR_INVALID_ARG, "%u triangles: a BVH leaf addresses 2^24", m->ntris);
    build_bvh(bt, 0, m->ntris, tree, order, (uint32_t)std::min(b->tune_bvh_leaf > 0 ? b->tune_bvh_leaf : 1, 255), b->tune_bvh_median != 0);
    std::vector<BvhNode>  nodes;                         // 8 layouts of tree.size() nodes each
    for (int octant = 0; octant < 8; ++octant) thread_bvh(tree, 0, octant, nodes, nodes.size());
    std::vector<TriDev> tris(order.size());
    for (size_t k = 0; k < order.size(); ++k) {
        const uint32_t  prim = bt[order[k]].prim;
        const uint32_t* idx = m->tri_idx + 3 * (size_t)prim;
        const float *v0 = m->verts_xyz + 3 * (size_t)idx[0], *v1 = m->verts_xyz + 3 * (size_t)idx[1], *v2 = m->verts_xyz + 3 * (size_t)idx[2];
        for (int a = 0; a < 3; ++a) { tris[k].v0[a] = v0[a]; tris[k].e1[a] = v1[a] - v0[a]; tris[k].e2[a] = v2[a] - v0[a]; }
        tris[k].prim = prim;
    }
    void* old[] = { b->d_nodes, b->d_tris, b->d_p2s, b->d_dissolve, b->d_diffuse };
    for (void* p : old) if (p) (void)hipFree(p);
    b->d_nodes = nullptr; b->d_tris = nullptr; b->d_p2s = nullptr; b->d_dissolve = nullptr; b->d_diffuse = nullptr;
    HIP_TRY(b, hipMalloc((void**)&b->d_nodes, nodes.size() * sizeof(BvhNode)));
    HIP_TRY(b, hipMalloc((void**)&b->d_tris, tris.size() * sizeof(TriDev)));
    HIP_TRY(b, hipMalloc((void**)&b->d_p2s, (size_t)m->ntris * sizeof(uint32_t)));
    HIP_TRY(b, hipMalloc((void**)&b->d_dissolve, (size_t)m->nshapes * sizeof(float)));
    HIP_TRY(b, hipMemcpy(b->d_nodes, nodes.data(), nodes.size() * sizeof(BvhNode), hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(b->d_tris, tris.data(), tris.size() * sizeof(TriDev), hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(b->d_p2s, m->prim_to_shape, (size_t)m->ntris * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_TRY(b, hipMemcpy(b->d_dissolve, m->shape_dissolve, (size_t)m->nshapes * sizeof(float), hipMemcpyHostToDevice));
    if (m->shape_diffuse) {
        HIP_TRY(b, hipMalloc((void**)&b->d_diffuse, (size_t)m->nshapes * 3 * sizeof(float)));
        HIP_TRY(b, hipMemcpy(b->d_diffuse, m->shape_diffuse, (size_t)m->nshapes * 3 * sizeof(float), hipMemcpyHostToDevice));
    }
    b->nnodes = (uint32_t)tree.size(); b->nshapes = m->nshapes;
    b->has_mesh = true;
    return VR_OK;
}

int vr_backend_render_row_set_device(vr_backend* b, const vr_options_pod* o, int integrator, uint32_t row_first,
                                     uint32_t row_stride, uint32_t row_count, void* d_rgb)
{
    if (!b) return VR_ERR_INVALID_ARG;
    return render_row_set(b, o, integrator, row_first, row_stride, row_count, (float*)d_rgb, false);
}

int vr_backend_render_row_set_device_noalloc(vr_backend* b, const vr_options_pod* o, int integrator, uint32_t row_first,
                                             uint32_t row_stride, uint32_t row_count, void* d_rgb)
{
    if (!b) return VR_ERR_INVALID_ARG;
    return render_row_set(b, o, integrator, row_first, row_stride, row_count, (float*)d_rgb, true);
}

int vr_backend_synchronize(vr_backend* b)
{
    if (!b) return VR_ERR_INVALID_ARG;
    int rc = use_device(b);
    if (rc) return rc;
    return collect(b);
}

int vr_backend_render_rows(vr_backend* b, const vr_options_pod* o, int integrator, uint32_t from_row, uint32_t to_row, float* pix)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (!o) return fail(b, VR_ERR_INVALID_ARG, "options == NULL");
    if (to_row < from_row || from_row < o->heightStartOffset) return fail(b, VR_ERR_INVALID_ARG, "bad row range [%u, %u)", from_row, to_row);
    if (to_row == from_row) return VR_OK;
    if (!pix) return fail(b, VR_ERR_INVALID_ARG, "pix == NULL");
    int rc = use_device(b);
    if (rc) return rc;
    const uint32_t rows = to_row - from_row;
    const size_t   nfl = (size_t)rows * o->width * 3;
    rc = ensure(b, b->d_rows, b->rows_cap, nfl);
    if (rc) return rc;
    rc = render_row_set(b, o, integrator, from_row - o->heightStartOffset, 1, rows, b->d_rows, false);
    if (rc) return rc;
    HIP_TRY(b, hipMemcpyAsync(pix, b->d_rows, nfl * sizeof(float), hipMemcpyDeviceToHost, b->stream));
    return collect(b);
}

int vr_backend_render(vr_backend* b, const vr_options_pod* o, int integrator, float* framebuffer)
{
    if (!b) return VR_ERR_INVALID_ARG;
    if (!o) return fail(b, VR_ERR_INVALID_ARG, "options == NULL");
    return vr_backend_render_rows(b, o, integrator, o->heightStartOffset, o->heightStartOffset + o->height, framebuffer);
}

int vr_backend_last_stats(vr_backend* b, vr_render_stats* stats, float* kernel_ms, uint32_t* launches)
{
    if (!b) return VR_ERR_INVALID_ARG;
    int rc = use_device(b);
    if (rc) return rc;
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    if (stats) *stats = b->stats;
    float total = 0.0f;
    for (size_t i = 0; i < b->passes_used; ++i) {
        float ms = 0.0f;
        HIP_TRY(b, hipEventElapsedTime(&ms, b->passes[i].e0, b->passes[i].e1));
        total += ms;
    }
    if (kernel_ms) *kernel_ms = total;
    if (launches) *launches = b->launches;
    return VR_OK;
}

int vr_backend_max_channel_device(vr_backend* b, const void* d_rgb, uint64_t npixels, float* max_out)
{
    if (!b || !d_rgb || !max_out) return VR_ERR_INVALID_ARG;
    int rc = use_device(b);
    if (rc) return rc;
    HIP_TRY(b, hipMemsetAsync(b->d_maxbits, 0, sizeof(int), b->stream));
    if (npixels) {
        k_max_channel<<<grid_blocks(npixels * 3, 1024), 256, 0, b->stream>>>((const float*)d_rgb, (size_t)npixels * 3, b->d_maxbits);
        HIP_TRY(b, hipGetLastError());
    }
    int bits = 0;
    HIP_TRY(b, hipMemcpyAsync(&bits, b->d_maxbits, sizeof bits, hipMemcpyDeviceToHost, b->stream));
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    memcpy(max_out, &bits, sizeof bits);
    return VR_OK;
}

int vr_backend_tonemap_device(vr_backend* b, void* d_rgb, uint64_t npixels, float max_white_in, float* max_white_out)
{
    if (!b || !d_rgb) return VR_ERR_INVALID_ARG;
    float mw = max_white_in;
    if (mw < 0.0f) {
        int rc = vr_backend_max_channel_device(b, d_rgb, npixels, &mw);
        if (rc) return rc;
    } else {
        int rc = use_device(b);
        if (rc) return rc;
    }
    if (max_white_out) *max_white_out = mw;
    if (npixels) {
        k_tonemap<<<grid_blocks(npixels * 3, 2048), 256, 0, b->stream>>>((float*)d_rgb, (size_t)npixels * 3, mw * mw);
        HIP_TRY(b, hipGetLastError());
    }
    HIP_TRY(b, hipStreamSynchronize(b->stream));
    return VR_OK;
}

// ---- one frame over several GPUs (BaseIntegrator.cpp:143-181) ----------------------------------
} // extern "C"

struct vr_backend_group {
    std::vector<vr_backend*> dev;
    std::string              err;
    bool                     rendered = false;
    int                      frame_path = 0;             // how the last vr_group_render reached the caller's frame: 0 = copies into pageable memory,
                                                         // 1 = the frame page-locked for the call, 2 = through the group's pinned staging buffer
    float*                   h_stage = nullptr;          // pinned host memory (hipHostMalloc), one contiguous row block per device
    size_t                   h_stage_cap = 0;            // in floats
    uint32_t                 peers_enabled = 0, peers_refused = 0;   // of the devices other than the first: peer access to the first GPU
};

namespace {
thread_local std::string g_group_create_error;

int group_fail(vr_backend_group* g, int code, uint32_t index, const vr_backend* b)
{
    char head[64];
    snprintf(head, sizeof head, "device %u of the group (GPU %d): ", index, b ? b->device : -1);
    g->err = std::string(head) + (b ? b->err : std::string("no backend"));
    return code;
}
#define VR_GROUP_EACH(g, call)                                             \
    do {                                                                   \
        if (!(g)) return VR_ERR_INVALID_ARG;                               \
        for (uint32_t i_ = 0; i_ < (g)->dev.size(); ++i_) {                \
            vr_backend* b = (g)->dev[i_];                                  \
            const int   rc_ = (call);                                      \
            if (rc_) return group_fail((g), rc_, i_, b);                   \
        }                                                                  \
        return VR_OK;                                                      \
    } while (0)

// rows i, i+N, ... of the frame on device i, all devices launched before any is waited for; each device then copies
// its row block to the rows' places in the frame (host memory, or memory of the group's first GPU).
// A host frame is the caller's `new Vec3f[]` — pageable memory, into which an asynchronous 2-D copy degrades to a
// synchronous, staged one, device after device.  So the frame is page-locked for the duration of the call
// (hipHostRegister; silently not when that fails, e.g. for memory that is registered already — then the copies are
// staged as before).  VR_GROUP_NO_REGISTER=1 keeps the old behaviour for an A/B.
// Failures: once a device has been launched nothing returns before every launched device has been waited for — the
// caller may free the frame on error, and copies into it must not be in flight then.
int group_render(vr_backend_group* g, const vr_options_pod* o, int integrator, float* frame, bool frame_on_device)
{
    if (!g) return VR_ERR_INVALID_ARG;
    g->rendered = false;
    if (!o || !frame) { g->err = "options or framebuffer == NULL"; return VR_ERR_INVALID_ARG; }
    const uint32_t N = (uint32_t)g->dev.size(), H = o->height, W = o->width;
    std::vector<uint32_t> count(N, 0);
    for (uint32_t i = 0; i < N; ++i) {
        vr_backend* b = g->dev[i];
        count[i] = i < H ? (H - i + N - 1) / N : 0;
        int rc = use_device(b);
        if (!rc && count[i]) rc = ensure(b, b->d_rows, b->rows_cap, (size_t)count[i] * W * 3);
        if (rc) return group_fail(g, rc, i, b);
    }
    const size_t row_bytes = (size_t)W * 3 * sizeof(float);
    // the way to the caller's host frame: page-lock it for the call; where the platform refuses that (hipHostRegister returns
    // "invalid argument" for any host pointer on some boxes), every device copies its row block — contiguous, asynchronous,
    // side by side — into the group's pinned staging buffer and the host moves the rows to their places as each device
    // finishes; the last resort is the strided copy into pageable memory
    bool registered = false, staged = false;
    std::vector<size_t> stage_off(N, 0);
    if (!frame_on_device && H && W) {
        const char* off = getenv("VR_GROUP_NO_REGISTER");
        const bool  plain = off && off[0] == '1';
        if (!plain) {
            registered = hipHostRegister(frame, (size_t)H * row_bytes, hipHostRegisterDefault) == hipSuccess;
            if (!registered) (void)hipGetLastError();
        }
        if (!registered && !plain) {
            const size_t need = (size_t)H * W * 3;
            if (g->h_stage_cap < need) {
                if (g->h_stage) (void)hipHostFree(g->h_stage);
                g->h_stage = nullptr; g->h_stage_cap = 0;
                if (hipHostMalloc((void**)&g->h_stage, need * sizeof(float), hipHostMallocDefault) == hipSuccess) g->h_stage_cap = need;
                else { g->h_stage = nullptr; (void)hipGetLastError(); }
            }
            staged = g->h_stage != nullptr;
            size_t o2 = 0;
            for (uint32_t i = 0; i < N; ++i) { stage_off[i] = o2; o2 += (size_t)count[i] * W * 3; }
        }
    }
    g->frame_path = registered ? 1 : (staged ? 2 : 0);
    int      first_rc = VR_OK;
    uint32_t launched = 0;                      // devices [0, launched) have work queued
    for (uint32_t i = 0; i < N && !first_rc; ++i) {
        vr_backend* b = g->dev[i];
        b->stats = vr_render_stats{}; b->passes_used = 0; b->launches = 0; b->last_counters = nullptr;
        int rc;
        if (!count[i]) rc = validate(b, o, integrator);      // a device without rows still reports a bad call
        else rc = render_row_set(b, o, integrator, i, N, count[i], b->d_rows, false);
        if (rc) first_rc = group_fail(g, rc, i, b);
        else launched = i + 1;
    }
    for (uint32_t i = 0; i < launched && !first_rc; ++i) {
        vr_backend* b = g->dev[i];
        if (!count[i]) continue;
        int rc = use_device(b);
        if (rc) { first_rc = group_fail(g, rc, i, b); break; }
        const hipError_t e = staged
            ? hipMemcpyAsync(g->h_stage + stage_off[i], b->d_rows, (size_t)count[i] * row_bytes, hipMemcpyDeviceToHost, b->stream)
            : hipMemcpy2DAsync((char*)frame + (size_t)i * row_bytes, (size_t)N * row_bytes, b->d_rows, row_bytes, row_bytes,
                               count[i], frame_on_device ? hipMemcpyDefault : hipMemcpyDeviceToHost, b->stream);
        if (e != hipSuccess) {
            fail(b, VR_ERR_HIP, "gather of the row block: %s", hipGetErrorString(e));
            first_rc = group_fail(g, VR_ERR_HIP, i, b);
        }
    }
    for (uint32_t i = 0; i < launched; ++i) {   // wait for every launched device, also after a failure
        vr_backend* b = g->dev[i];
        int rc = use_device(b);
        if (!rc) rc = count[i] ? collect(b) : VR_OK;
        if (rc && !first_rc) first_rc = group_fail(g, rc, i, b);
        if (staged && !first_rc && count[i]) {
            const float* src = g->h_stage + stage_off[i];
            for (uint32_t k = 0; k < count[i]; ++k)
                memcpy((char*)frame + ((size_t)i + (size_t)k * N) * row_bytes, (const char*)src + (size_t)k * row_bytes, row_bytes);
        }
    }
    if (registered && hipHostUnregister(frame) != hipSuccess) (void)hipGetLastError();
    g->rendered = first_rc == VR_OK;
    return first_rc;
}
} // namespace

extern "C" {

int vr_group_create(const int* ordinals, uint32_t ndevices, vr_backend_group** out)
{
    if (!out) { g_group_create_error = "out == NULL"; return VR_ERR_INVALID_ARG; }
    *out = nullptr;
    if (!ordinals || ndevices == 0 || ndevices > 64) { g_group_create_error = "device list empty, NULL or longer than 64"; return VR_ERR_INVALID_ARG; }
    vr_backend_group* g = new vr_backend_group;
    for (uint32_t i = 0; i < ndevices; ++i) {
        vr_backend* b = nullptr;
        const int   rc = vr_backend_create(ordinals[i], &b);
        if (rc) {
            g_group_create_error = "device " + std::to_string(i) + " of the list: " + vr_backend_last_error(nullptr);
            vr_group_destroy(g);
            return rc;
        }
        g->dev.push_back(b);
    }
    // the first GPU's memory is the target of vr_group_render_device's gather: let the others write to it
    for (uint32_t i = 1; i < ndevices; ++i) {
        if (g->dev[i]->device == g->dev[0]->device) continue;
        int can = 0;
        bool ok = false;
        if (hipDeviceCanAccessPeer(&can, g->dev[i]->device, g->dev[0]->device) == hipSuccess && can) {
            (void)hipSetDevice(g->dev[i]->device);
            const hipError_t e = hipDeviceEnablePeerAccess(g->dev[0]->device, 0);
            ok = e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled;
        }
        (void)hipGetLastError();
        if (ok) g->peers_enabled++;
        else {
            // not an error: vr_group_render_device's copies from that GPU then go through the host; counted so that a slow
            // gather can be diagnosed (vr_group_gather_info)
            g->peers_refused++;
        }
    }
    *out = g;
    return VR_OK;
}

void vr_group_destroy(vr_backend_group* g)
{
    if (!g) return;
    if (g->h_stage) (void)hipHostFree(g->h_stage);
    for (vr_backend* b : g->dev) vr_backend_destroy(b);
    delete g;
}

uint32_t    vr_group_size(const vr_backend_group* g) { return g ? (uint32_t)g->dev.size() : 0u; }
vr_backend* vr_group_backend(vr_backend_group* g, uint32_t index) { return g && index < g->dev.size() ? g->dev[index] : nullptr; }
const char* vr_group_last_error(const vr_backend_group* g) { return g ? g->err.c_str() : g_group_create_error.c_str(); }

int vr_group_set_density(vr_backend_group* g, const vr_grid_desc* density, const vr_grid_desc* temperature) { VR_GROUP_EACH(g, vr_backend_set_density(b, density, temperature)); }
int vr_group_generate_cloud(vr_backend_group* g, int n, float radius_voxels) { VR_GROUP_EACH(g, vr_backend_generate_cloud(b, n, radius_voxels)); }
int vr_group_set_density_map(vr_backend_group* g, float voxel_size, const float origin_xyz[3]) { VR_GROUP_EACH(g, vr_backend_set_density_map(b, voxel_size, origin_xyz)); }
int vr_group_set_mesh(vr_backend_group* g, const vr_mesh_desc* mesh) { VR_GROUP_EACH(g, vr_backend_set_mesh(b, mesh)); }
int vr_group_set_mode(vr_backend_group* g, int mode) { VR_GROUP_EACH(g, vr_backend_set_mode(b, mode)); }
int vr_group_set_seed(vr_backend_group* g, uint64_t seed) { VR_GROUP_EACH(g, vr_backend_set_seed(b, seed)); }

int vr_group_render(vr_backend_group* g, const vr_options_pod* o, int integrator, float* framebuffer_rgb)
{
    return group_render(g, o, integrator, framebuffer_rgb, false);
}
int vr_group_render_device(vr_backend_group* g, const vr_options_pod* o, int integrator, void* d_rgb)
{
    return group_render(g, o, integrator, (float*)d_rgb, true);
}

int vr_group_gather_info(const vr_backend_group* g, uint32_t out[3])
{
    if (!g || !out) return VR_ERR_INVALID_ARG;
    out[0] = g->peers_enabled; out[1] = g->peers_refused; out[2] = (uint32_t)g->frame_path;
    return VR_OK;
}

int vr_group_last_stats(vr_backend_group* g, vr_render_stats* stats, float* kernel_ms, uint32_t* kernel_launches)
{
    if (!g) return VR_ERR_INVALID_ARG;
    vr_render_stats sum{};
    float           slowest = 0.0f;
    uint32_t        launches = 0;
    for (uint32_t i = 0; i < g->dev.size(); ++i) {
        vr_render_stats st{};
        float           ms = 0.0f;
        uint32_t        n = 0;
        const int rc = vr_backend_last_stats(g->dev[i], &st, &ms, &n);
        if (rc) return group_fail(g, rc, i, g->dev[i]);
        uint64_t*       a = (uint64_t*)&sum;
        const uint64_t* c = (const uint64_t*)&st;
        for (size_t k = 0; k < sizeof(vr_render_stats) / sizeof(uint64_t); ++k) a[k] += c[k];
        slowest = std::max(slowest, ms);
        launches += n;
    }
    if (stats) *stats = sum;
    if (kernel_ms) *kernel_ms = slowest;
    if (kernel_launches) *kernel_launches = launches;
    return VR_OK;
}

int vr_rgb_to_u8(const float* rgb, uint32_t width, uint32_t height, uint8_t* out)
{
    if (!rgb || !out) return VR_ERR_INVALID_ARG;
    const size_t n = (size_t)width * height * 3;
    for (size_t i = 0; i < n; ++i) {
        const float v = rgb[i];
        const float m = (v < 1.0f) ? v : 1.0f;        // Utils::clamp(0, 1, v)              BaseIntegrator.cpp:21-23
        const float c = (0.0f < m) ? m : 0.0f;
        out[i] = (uint8_t)(255 * c);
    }
    return VR_OK;
}

} // extern "C"
