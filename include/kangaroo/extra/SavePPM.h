// SavePPM.h -- SavePXM / LoadPXM for roo::Volume / roo::BoundedVolume (reference
// include/kangaroo/extra/SavePPM.h:42-195).  On-disk layout: two text lines with the bounding box
// (BoundedVolume only, operator<< float), "P5\n", "w h d\n", "255\n", then d*h packed rows of w*sizeof(T)
// bytes.  Device volumes go through one pitched copy into a packed host buffer (kfx_memcpy_2d).
#pragma once

#include <fstream>
#include <string>
#include <vector>

#include <kangaroo/BoundedVolume.h>
#include <kangaroo/Volume.h>

namespace pxm_detail
{
// rows of a (possibly pitched) volume -> packed bytes, from either memory space
template<typename T, typename Target, typename Management>
inline bool PackedRows(std::vector<char>& out, const roo::Volume<T,Target,Management>& vol)
{
    const size_t rowbytes = vol.w * sizeof(T);
    out.resize(rowbytes * vol.h * vol.d);
    const int kind = roo::TargetCopyKind<roo::TargetHost,Target>();
    for (size_t z = 0; z < vol.d; ++z)
        if (kfx_memcpy_2d(out.data() + z * vol.h * rowbytes, rowbytes, (const char*)vol.ptr + z * vol.img_pitch, vol.pitch, rowbytes, vol.h, kind, 0) != 0)
            return false;
    return true;
}
}

template<typename T, typename Target, typename Management>
void SavePXM(std::ofstream& bFile, const roo::Volume<T,Target,Management>& vol, std::string ppm_type = "P5", int num_colors = 255)
{
    std::vector<char> rows;
    pxm_detail::PackedRows(rows, vol);
    bFile << ppm_type << std::endl;
    bFile << vol.w << " " << vol.h << " " << vol.d << '\n';
    bFile << num_colors << '\n';
    bFile.write(rows.data(), (std::streamsize)rows.size());
    bFile.close();
}

template<typename T, typename Target, typename Management>
void SavePXM(const std::string filename, const roo::Volume<T,Target,Management>& vol, std::string ppm_type = "P5", int num_colors = 255)
{
    std::ofstream bFile(filename.c_str(), std::ios::out | std::ios::binary);
    SavePXM(bFile, vol, ppm_type, num_colors);
}

template<typename T, typename Target, typename Management>
void SavePXM(const std::string filename, const roo::BoundedVolume<T,Target,Management>& vol, std::string ppm_type = "P5", int num_colors = 255)
{
    std::ofstream bFile(filename.c_str(), std::ios::out | std::ios::binary);
    bFile << vol.bbox.boxmin.x << " " << vol.bbox.boxmin.y << " " << vol.bbox.boxmin.z << std::endl;
    bFile << vol.bbox.boxmax.x << " " << vol.bbox.boxmax.y << " " << vol.bbox.boxmax.z << std::endl;
    SavePXM(bFile, static_cast<const roo::Volume<T,Target,Management>&>(vol), ppm_type, num_colors);
}

// Replaces the contents of an owning volume with the file's (any previous allocation is released).
template<typename T, typename Target>
bool LoadPXM(std::ifstream& bFile, roo::Volume<T,Target,roo::Manage>& vol)
{
    std::string ppm_type = "";
    int num_colors = 0, w = 0, h = 0, d = 0;
    bFile >> ppm_type;
    bFile >> w;
    bFile >> h;
    bFile >> d;
    bFile >> num_colors;
    bFile.ignore(1, '\n');
    bool success = !bFile.fail() && w > 0 && h > 0 && d > 0;
    if (success) {
        const size_t rowbytes = (size_t)w * sizeof(T);
        std::vector<char> rows(rowbytes * h * d);
        bFile.read(rows.data(), (std::streamsize)rows.size());
        success = !bFile.fail();
        if (success) {
            roo::Manage::Cleanup<T,Target>(vol.ptr);
            Target::template AllocatePitchedMem<T>(&vol.ptr, &vol.pitch, &vol.img_pitch, w, h, d);
            vol.w = w; vol.h = h; vol.d = d;
            const int kind = roo::TargetCopyKind<Target,roo::TargetHost>();
            for (int z = 0; z < d && success; ++z)
                success = kfx_memcpy_2d((char*)vol.ptr + (size_t)z * vol.img_pitch, vol.pitch, rows.data() + (size_t)z * h * rowbytes, rowbytes, rowbytes, h, kind, 0) == 0;
        }
    }
    bFile.close();
    return success;
}

template<typename T, typename Target>
bool LoadPXM(const std::string filename, roo::Volume<T,Target,roo::Manage>& vol)
{
    std::ifstream bFile(filename.c_str(), std::ios::in | std::ios::binary);
    return LoadPXM(bFile, vol);
}

template<typename T, typename Target>
bool LoadPXM(const std::string filename, roo::BoundedVolume<T,Target,roo::Manage>& vol)
{
    std::ifstream bFile(filename.c_str(), std::ios::in | std::ios::binary);
    bFile >> vol.bbox.boxmin.x;
    bFile >> vol.bbox.boxmin.y;
    bFile >> vol.bbox.boxmin.z;
    bFile >> vol.bbox.boxmax.x;
    bFile >> vol.bbox.boxmax.y;
    bFile >> vol.bbox.boxmax.z;
    bFile.ignore(1, '\n');
    return LoadPXM(bFile, static_cast<roo::Volume<T,Target,roo::Manage>&>(vol));
}
