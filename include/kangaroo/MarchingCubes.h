// MarchingCubes.h -- roo::SaveMesh (reference include/kangaroo/MarchingCubes.h:205-262): iso-surface of a device
// BoundedVolume<SDF_t>, optionally coloured from a BoundedVolume<float>, written as <filename>.ply.
// The reference marches the cubes on the host after copying the volume back and exports through Assimp; here the
// extraction runs on the GPU (kfx_mc_count -> prefix sum -> kfx_mc_emit) and only the finished vertex arrays are
// copied.  Vertex order, positions, normals and colours follow the reference's loop nest and expressions; the
// case tables are this repo's own derivation (scripts/gen_mc_tables.py, same boundary loops and winding as the
// classic tables in all 256 cases).  PLY layout: x y z nx ny nz [red green blue alpha] floats, one face per three
// consecutive vertices, binary little endian.
#pragma once

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Sdf.h>
#include <kangaroo/launch_utils.h>

namespace roo
{

namespace mesh_detail
{
inline void* DeviceBytes(size_t n)
{
    void* p = nullptr;
    size_t pitch = 0;
    GpuCheckStatus(kfx_alloc_pitched(&p, &pitch, n ? n : 1, 1));
    return p;
}
}

// Returns the number of triangles written; counts / offsets live on the device, the prefix sum of the per-cube
// counts (at most 5 each) is taken on the host over one byte per cube.
template<typename Manage1, typename Manage2>
inline size_t SaveMesh(std::string filename, BoundedVolume<SDF_t,TargetDevice,Manage1>& vol, BoundedVolume<float,TargetDevice,Manage2>* volColor)
{
    const size_t ncubes = (vol.w - 1) * (vol.h - 1) * (vol.d - 1);
    unsigned char* dcounts = (unsigned char*)mesh_detail::DeviceBytes(ncubes);
    GpuCheckStatus(kfx_mc_count(vol.abi(), dcounts, 0));
    std::vector<unsigned char> counts(ncubes);
    GpuCheckStatus(kfx_memcpy_2d(counts.data(), ncubes, dcounts, ncubes, ncubes, 1, 2, 0));
    std::vector<long long> active;     // cubes with triangles, in emission order
    std::vector<unsigned> tri_offset;  // triangles emitted before each of them
    size_t ntri = 0;
    for (size_t i = 0; i < ncubes; ++i)
        if (counts[i]) { active.push_back((long long)i); tri_offset.push_back((unsigned)ntri); ntri += counts[i]; }
    const size_t na = active.size();
    long long* dactive = (long long*)mesh_detail::DeviceBytes(na * 8);
    unsigned* doffsets = (unsigned*)mesh_detail::DeviceBytes(na * 4);
    if (na) {
        GpuCheckStatus(kfx_memcpy_2d(dactive, na * 8, active.data(), na * 8, na * 8, 1, 1, 0));
        GpuCheckStatus(kfx_memcpy_2d(doffsets, na * 4, tri_offset.data(), na * 4, na * 4, 1, 1, 0));
    }
    const bool color = volColor && volColor->IsValid();
    const size_t nv = 3 * ntri;
    float* dv = (float*)mesh_detail::DeviceBytes(nv * 12);
    float* dn = (float*)mesh_detail::DeviceBytes(nv * 12);
    float* dc = color ? (float*)mesh_detail::DeviceBytes(nv * 16) : nullptr;
    if (ntri) GpuCheckStatus(kfx_mc_emit(vol.abi(), color ? volColor->abi() : nullptr, dactive, doffsets, (long long)na, dv, dn, dc, 0));
    std::vector<float> v(nv * 3), n(nv * 3), c(color ? nv * 4 : 0);
    if (nv) {
        GpuCheckStatus(kfx_memcpy_2d(v.data(), nv * 12, dv, nv * 12, nv * 12, 1, 2, 0));
        GpuCheckStatus(kfx_memcpy_2d(n.data(), nv * 12, dn, nv * 12, nv * 12, 1, 2, 0));
        if (color) GpuCheckStatus(kfx_memcpy_2d(c.data(), nv * 16, dc, nv * 16, nv * 16, 1, 2, 0));
    }
    kfx_free(dcounts); kfx_free(dactive); kfx_free(doffsets); kfx_free(dv); kfx_free(dn);
    if (dc) kfx_free(dc);

    FILE* f = fopen((filename + ".ply").c_str(), "wb");
    if (!f) return 0;
    fprintf(f, "ply\nformat binary_little_endian 1.0\ncomment kangaroo_amd marching cubes\nelement vertex %zu\n", nv);
    fprintf(f, "property float x\nproperty float y\nproperty float z\nproperty float nx\nproperty float ny\nproperty float nz\n");
    if (color) fprintf(f, "property float red\nproperty float green\nproperty float blue\nproperty float alpha\n");
    fprintf(f, "element face %zu\nproperty list uchar uint vertex_indices\nend_header\n", ntri);
    for (size_t i = 0; i < nv; ++i) {
        fwrite(&v[i * 3], 4, 3, f);
        fwrite(&n[i * 3], 4, 3, f);
        if (color) fwrite(&c[i * 4], 4, 4, f);
    }
    for (size_t t = 0; t < ntri; ++t) {
        const unsigned char k = 3;
        const uint32_t idx[3] = {(uint32_t)(3 * t), (uint32_t)(3 * t + 1), (uint32_t)(3 * t + 2)};
        fwrite(&k, 1, 1, f);
        fwrite(idx, 4, 3, f);
    }
    fclose(f);
    return ntri;
}

template<typename Manage>
inline size_t SaveMesh(std::string filename, BoundedVolume<SDF_t,TargetDevice,Manage>& vol)
{
    return SaveMesh<Manage,Manage>(filename, vol, nullptr);
}

template<typename Manage1, typename Manage2>
inline size_t SaveMesh(std::string filename, BoundedVolume<SDF_t,TargetDevice,Manage1>& vol, BoundedVolume<float,TargetDevice,Manage2>& volColor)
{
    return SaveMesh<Manage1,Manage2>(filename, vol, &volColor);
}

}
