// mesh.hip -- iso-surface extraction from the TSDF (roo::SaveMesh's marching cubes) for gfx950.
// SURVEY.md 8(f) row f-4.
//
// Reference behaviour: include/kangaroo/MarchingCubes.h:43-143 (vMarchCube: corner values, case index,
// edge vertices by linear interpolation, normals from GetUnitsBackwardDiffDxDyDz, grey colour from the
// colour volume) inside the loop nest of SaveMesh (:226-232: x outer, y, z inner, three fresh vertices per
// triangle).  The reference does this on the host, one cube at a time, after copying the volume back.
//
// Here the volume never leaves HBM.  Two passes over the (w-1)(h-1)(d-1) cubes:
//   k_mc_count  one thread per cube, x fastest (coalesced corner rows): number of triangles of the cube's case,
//               written at the cube's position in the REFERENCE's emission order ((x*(h-1) + y)*(d-1) + z)
//               through an LDS transpose (the order is z-fastest);
//   (host side: exclusive prefix sum of the counts = every cube's output slot; the non-zero positions = the list
//    of active cubes, already in emission order -- torch.cumsum / torch.nonzero, or a host loop in the C++ header)
//   k_mc_emit   one thread per active cube: edge vertices, normals, colours, written to the cube's slots.  The
//               vertex / normal / colour arithmetic keeps the reference's expressions (double division in
//               fGetOffset, multiply-by-reciprocal normalisation), so the output arrays are bit-identical to the
//               CPU oracle's and arrive in the reference's order.
// The case tables (mc_tables.inc) are derived by scripts/gen_mc_tables.py from the cube's topology; their boundary
// loops and winding equal the classic tables' in all 256 cases (tests/test_mesh_cpu.py).
#include "kfx_device.h"
#include "sampling.h"

namespace kfx {

#include "mc_tables.inc"

__constant__ unsigned char c_num_tris[256];
__constant__ unsigned short c_edge_mask[256];
__constant__ signed char c_tris[256][15];

struct MeshParams {
    VolView vol;
    V3 size, dims1, hi2, voxel;  // members trilinear<>() / gradient<>() expect
    V3 inv_size;
    int fastdiv, off32;
    int cx, cy, cz;              // cubes per axis = dims - 1
};

// corner i of the cube at (x, y, z): offsets (0,0,0) (1,0,0) (1,1,0) (0,1,0) (0,0,1) (1,0,1) (1,1,1) (0,1,1)
__device__ __forceinline__ int corner_dx(int i) { return ((i + 1) >> 1) & 1; }
__device__ __forceinline__ int corner_dy(int i) { return (i >> 1) & 1; }
__device__ __forceinline__ int corner_dz(int i) { return i >> 2; }

// values at the 8 corners and the case index; false if a corner is not finite (MarchingCubes.h:58-74)
__device__ __forceinline__ bool cube_case(const MeshParams& p, int x, int y, int z, float v[8], int& flag)
{
    const unsigned char* r00 = rowp(p.vol, y, z);
    const unsigned char* r10 = rowp(p.vol, y + 1, z);
    const unsigned char* r01 = rowp(p.vol, y, z + 1);
    const unsigned char* r11 = rowp(p.vol, y + 1, z + 1);
    const float2 a = RayF32::pair(r00, x), b = RayF32::pair(r10, x), c = RayF32::pair(r01, x), d = RayF32::pair(r11, x);
    v[0] = a.x; v[1] = a.y; v[2] = b.y; v[3] = b.x;
    v[4] = c.x; v[5] = c.y; v[6] = d.y; v[7] = d.x;
    bool finite = true;
    flag = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        finite = finite && isfinite(v[i]);
        if (v[i] <= 0.0f) flag |= 1 << i;
    }
    return finite;
}

// Workgroup = 64 cubes along x, 16 along z, one y.  Corner rows are read x-fastest (coalesced); the counts go
// through an LDS tile so that each x-row's 16 z-consecutive bytes leave as one segment (the output order is z-fastest).
__global__ __launch_bounds__(256) void k_mc_count(const MeshParams p, unsigned char* __restrict__ counts)
{
    __shared__ unsigned char tile[64][17];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int x0 = blockIdx.x * 64, y = blockIdx.y, z0 = blockIdx.z * 16;
    const int x = x0 + lane;
    for (int zz = wv; zz < 16; zz += 4) {
        const int z = z0 + zz;
        unsigned char n = 0;
        if (x < p.cx && z < p.cz) {
            float v[8];
            int flag;
            if (cube_case(p, x, y, z, v, flag)) n = c_num_tris[flag];
        }
        tile[lane][zz] = n;
    }
    __syncthreads();
    const int xr = threadIdx.x >> 2, q = threadIdx.x & 3;
    if (x0 + xr < p.cx) {
        unsigned char* dst = counts + ((size_t)(x0 + xr) * p.cy + y) * p.cz;
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int z = z0 + q * 4 + b;
            if (z < p.cz) dst[z] = tile[xr][q * 4 + b];
        }
    }
}

// One thread per ACTIVE cube (cube_index lists the cubes with triangles in emission order; tri_offset their first
// triangle): dense lanes instead of the ~1 % active lanes of a thread-per-cube sweep.
__global__ __launch_bounds__(128) void k_mc_emit(const MeshParams p, const ColorGeom cv, const int has_color,
                                                 const long long* __restrict__ cube_index, const unsigned* __restrict__ tri_offset,
                                                 const long long n_active, float* __restrict__ verts,
                                                 float* __restrict__ norms, float* __restrict__ colors)
{
    const long long t_id = (long long)blockIdx.x * 128 + threadIdx.x;
    if (t_id >= n_active) return;
    const long long ci = cube_index[t_id];
    const int z = (int)(ci % p.cz);
    const int y = (int)((ci / p.cz) % p.cy);
    const int x = (int)(ci / ((long long)p.cz * p.cy));
    float v[8];
    int flag;
    if (!cube_case(p, x, y, z, v, flag)) return;
    const int ntri = c_num_tris[flag];
    if (ntri == 0) return;
    const unsigned mask = c_edge_mask[flag];
    // VoxelPositionInUnits(x,y,z) and VoxelSizeUnits() (BoundedVolume.h:115-125, :67-76)
    const V3 p0 = v3(p.vol.bmin.x + p.size.x * (float)x / p.dims1.x, p.vol.bmin.y + p.size.y * (float)y / p.dims1.y,
                     p.vol.bmin.z + p.size.z * (float)z / p.dims1.z);
    V3 ev[12], en[12];
    float ec[12];
#pragma unroll
    for (int e = 0; e < 12; ++e) {
        if (!(mask & (1u << e))) continue;
        // edges 0-3: bottom ring, 4-7: top ring, 8-11: verticals (MarchingCubes tables' numbering)
        const int c0 = e < 8 ? e : e - 8, c1 = e < 4 ? (e + 1) & 3 : (e < 8 ? 4 + ((e + 1) & 3) : e - 4);
        // fGetOffset (MarchingCubes.h:25-32): the difference is a float, the quotient a double
        const double delta = (double)(v[c1] - v[c0]);
        const float off = delta == 0.0 ? 0.5f : (float)((double)(0.0f - v[c0]) / delta);
        const float ox = (float)corner_dx(c0), oy = (float)corner_dy(c0), oz = (float)corner_dz(c0);
        const float dx = (float)(corner_dx(c1) - corner_dx(c0)), dy = (float)(corner_dy(c1) - corner_dy(c0)),
                    dz = (float)(corner_dz(c1) - corner_dz(c0));
        const V3 pos = v3(p0.x + (ox + off * dx) * p.voxel.x, p0.y + (oy + off * dy) * p.voxel.y, p0.z + (oz + off * dz) * p.voxel.z);
        ev[e] = pos;
        const V3 deriv = gradient<RayF32>(p, pos);
        V3 n = div_s(deriv, length(deriv));
        if (!isfinite(n.x) || !isfinite(n.y) || !isfinite(n.z)) n = v3(0.f, 0.f, 0.f);
        en[e] = n;
        ec[e] = has_color ? trilinear<RayC32>(cv, pos) : 0.f;
    }
    size_t o = (size_t)tri_offset[t_id] * 3; // first output vertex of this cube
    for (int t = 0; t < ntri * 3; ++t, ++o) {
        const int e = c_tris[flag][t];
        V3 P = v3(0.f, 0.f, 0.f), N = P;
        float C = 0.f;
#pragma unroll
        for (int k = 0; k < 12; ++k)   // select without dynamic register indexing
            if (k == e) { P = ev[k]; N = en[k]; C = ec[k]; }
        verts[o * 3 + 0] = P.x; verts[o * 3 + 1] = P.y; verts[o * 3 + 2] = P.z;
        norms[o * 3 + 0] = N.x; norms[o * 3 + 1] = N.y; norms[o * 3 + 2] = N.z;
        if (has_color) { // ConvertPixel<float3,float>(c) = (c,c,c); aiColor4D(c, c, c, 1)
            colors[o * 4 + 0] = C; colors[o * 4 + 1] = C; colors[o * 4 + 2] = C; colors[o * 4 + 3] = 1.0f;
        }
    }
}

static bool g_tables_loaded[64] = {};

static int load_tables()
{
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (g_tables_loaded[dev]) return 0;
    hipError_t e = hipMemcpyToSymbol(HIP_SYMBOL(c_num_tris), MC_NUM_TRIS, sizeof(MC_NUM_TRIS));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_edge_mask), MC_EDGE_MASK, sizeof(MC_EDGE_MASK));
    if (e == hipSuccess) e = hipMemcpyToSymbol(HIP_SYMBOL(c_tris), MC_TRIS, sizeof(MC_TRIS));
    if (e != hipSuccess) return set_error((int)e, hipGetErrorString(e));
    g_tables_loaded[dev] = true;
    return 0;
}

static int mesh_params(MeshParams& p, const kfx_volume* vol)
{
    if (!vol || !vol->ptr) return set_error(KFX_E_NULL, "SaveMesh: null volume");
    if (vol->w < 3 || vol->h < 3 || vol->d < 3 || vol->w > 65535 || vol->h > 65535 || vol->d > 65535)
        return set_error(KFX_E_SHAPE, "SaveMesh: volume dimensions");
    if (vol->pitch < vol->w * 8 || vol->img_pitch < vol->pitch * (vol->h - 1) + vol->w * 8) return set_error(KFX_E_SHAPE, "SaveMesh: volume pitch");
    if (((uintptr_t)vol->ptr | vol->pitch | vol->img_pitch) & 7) return set_error(KFX_E_ALIGN, "SaveMesh: alignment");
    set_geometry(p, vol);
    set_voxel_size(p, vol);
    p.cx = (int)vol->w - 1; p.cy = (int)vol->h - 1; p.cz = (int)vol->d - 1;
    return 0;
}

} // namespace kfx

using namespace kfx;

extern "C" int kfx_mc_count(const kfx_volume* vol, unsigned char* counts, kfx_stream stream)
{
    MeshParams p;
    if (int e = mesh_params(p, vol)) return e;
    if (!counts) return set_error(KFX_E_NULL, "SaveMesh: null counts");
    if (int e = load_tables()) return e;
    dim3 grid(ceil_div(p.cx, 64), p.cy, ceil_div(p.cz, 16));
    hipLaunchKernelGGL(k_mc_count, grid, dim3(256), 0, (hipStream_t)stream, p, counts);
    return check_launch("kfx_mc_count");
}

extern "C" int kfx_mc_emit(const kfx_volume* vol, const kfx_volume* colorvol, const long long* cube_index, const unsigned* tri_offset,
                           long long n_active, float* verts, float* norms, float* colors, kfx_stream stream)
{
    MeshParams p;
    if (int e = mesh_params(p, vol)) return e;
    if (n_active <= 0) return 0;
    if (!cube_index || !tri_offset || !verts || !norms) return set_error(KFX_E_NULL, "SaveMesh: null output");
    if (int e = load_tables()) return e;
    ColorGeom cv{};
    // the reference samples the colour volume only when it IsValid(): every dimension >= 8 (BoundedVolume.h:84-87)
    const int has_color = colorvol && colorvol->ptr && colors && colorvol->w >= 8 && colorvol->h >= 8 && colorvol->d >= 8;
    if (has_color) {
        if (colorvol->pitch < colorvol->w * 4 || colorvol->img_pitch < colorvol->pitch * (colorvol->h - 1) + colorvol->w * 4)
            return set_error(KFX_E_SHAPE, "SaveMesh: colour volume pitch");
        if (((uintptr_t)colorvol->ptr | colorvol->pitch | colorvol->img_pitch) & 3) return set_error(KFX_E_ALIGN, "SaveMesh: colour volume alignment");
        set_geometry(cv, colorvol);
    }
    if (n_active > 0x7fffffffLL * 128) return set_error(KFX_E_RANGE, "SaveMesh: too many active cubes");
    hipLaunchKernelGGL(k_mc_emit, dim3((unsigned)((n_active + 127) / 128)), dim3(128), 0, (hipStream_t)stream, p, cv, has_color,
                       cube_index, tri_offset, n_active, verts, norms, colors);
    return check_launch("kfx_mc_emit");
}
