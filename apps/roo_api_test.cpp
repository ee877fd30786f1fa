// roo_api_test.cpp -- exercises the roo:: containers and operators from C++ the way a Kangaroo
// application does, and checks device results against the containers' own host-side methods
// (the same header code, run on TargetHost copies).  Exit code 0 = all checks passed.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include <kangaroo/kangaroo.h>
#include <kangaroo/SdfSummary.h>
#include <kangaroo/extra/SavePPM.h>

using namespace roo;

static int g_fail = 0;
#define CHECK(cond)                                                                   \
    do {                                                                              \
        if (!(cond)) { ++g_fail; fprintf(stderr, "CHECK failed %s:%d: %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

static bool same(float a, float b) { return (a == b) || (std::isnan(a) && std::isnan(b)); }

int main()
{
    // ---- layouts the C ABI relies on ----
    CHECK(sizeof(Image<float>) == 32);
    CHECK(sizeof(Volume<SDF_t>) == 48);
    CHECK(sizeof(BoundedVolume<SDF_t>) == 72);
    CHECK(sizeof(SDF_t) == 8 && alignof(SDF_t) == 8);
    CHECK((sizeof(Mat<float,3,4>) == 48));
    CHECK(sizeof(ImageIntrinsics) == 16 && sizeof(BoundingBox) == 24);

    // ---- ownership policy: allocating a non-owning container throws ----
    bool threw = false;
    try { Image<float> bad(4, 4); } catch (const HipException&) { threw = true; }
    CHECK(threw);

    // ---- SDF_t running average ----
    SDF_t a(0.5f, 2.0f);
    a += SDF_t(NAN, 0.0f);                 // never-observed cell: untouched
    CHECK(a.val == 0.5f && a.w == 2.0f);
    a += SDF_t(-0.5f, 2.0f);
    CHECK(a.val == 0.0f && a.w == 4.0f);
    a.LimitWeight(3.0f);
    CHECK(a.w == 3.0f);

    if (kfx_device_count() < 1) {
        printf("roo_api_test: host-only checks %s (no HIP device)\n", g_fail ? "FAILED" : "passed");
        return g_fail ? 1 : 0;
    }

    const int N = 32, w = 64, h = 48;
    const ImageIntrinsics K(500, 500, w / 2, h / 2);
    BoundedVolume<SDF_t, TargetDevice, Manage> vol(N, N, N, make_float3(-1, -1, -1), make_float3(1, 1, 1));
    BoundedVolume<SDF_t, TargetHost, Manage> hvol(N, N, N, make_float3(-1, -1, -1), make_float3(1, 1, 1));
    CHECK(vol.pitch % 256 == 0 && vol.img_pitch == vol.pitch * N);

    // ---- SdfSphere on the device vs VoxelPositionInUnits on the host ----
    SdfReset(vol, NAN);
    SdfSphere(vol, make_float3(0, 0, 0), 0.9f);
    // device volume is pitched; copy row by row into the packed host volume
    CHECK(kfx_memcpy_2d(hvol.ptr, hvol.pitch, vol.ptr, vol.pitch, N * sizeof(SDF_t), (size_t)N * N, 2, 0) == 0);
    int bad = 0;
    for (int z = 0; z < N; ++z)
        for (int y = 0; y < N; ++y)
            for (int x = 0; x < N; ++x) {
                const float3 p = hvol.VoxelPositionInUnits(x, y, z);
                const float expect = length(p) - 0.9f;
                if (!(hvol(x, y, z).val == expect && hvol(x, y, z).w == 1.0f)) ++bad;
            }
    CHECK(bad == 0);

    // ---- persistence: SavePXM of the device volume, LoadPXM into a fresh one (extra/SavePPM.h) ----
    {
        const std::string path = "/tmp/roo_api_test_save.vol";
        SavePXM(path, vol);
        BoundedVolume<SDF_t, TargetDevice, Manage> vol2;
        CHECK(LoadPXM(path, vol2));
        CHECK(vol2.w == (size_t)N && vol2.h == (size_t)N && vol2.d == (size_t)N);
        CHECK(vol2.bbox.Min().x == -1.0f && vol2.bbox.Max().z == 1.0f);
        BoundedVolume<SDF_t, TargetHost, Manage> h2(N, N, N, make_float3(-1, -1, -1), make_float3(1, 1, 1));
        CHECK(kfx_memcpy_2d(h2.ptr, h2.pitch, vol2.ptr, vol2.pitch, N * sizeof(SDF_t), (size_t)N * N, 2, 0) == 0);
        int diff = 0;
        for (int z = 0; z < N; ++z)
            for (int y = 0; y < N; ++y)
                for (int x = 0; x < N; ++x)
                    if (!(h2(x, y, z).val == hvol(x, y, z).val && h2(x, y, z).w == hvol(x, y, z).w)) ++diff;
        CHECK(diff == 0);
        BoundedVolume<SDF_t, TargetDevice, Manage> missing;
        CHECK(!LoadPXM("/tmp/roo_api_test_no_such_file.vol", missing));
        remove(path.c_str());
    }

    // ---- mesh extraction: SaveMesh of the sphere (MarchingCubes.h) -- triangle count, vertices on the sphere ----
    {
        const size_t ntri = SaveMesh("/tmp/roo_api_test_mesh", vol);
        CHECK(ntri > 1000);
        FILE* f = fopen("/tmp/roo_api_test_mesh.ply", "rb");
        CHECK(f != nullptr);
        if (f) {
            char line[256];
            size_t nv = 0;
            while (fgets(line, sizeof line, f) && strncmp(line, "end_header", 10) != 0)
                if (sscanf(line, "element vertex %zu", &nv) == 1) {}
            CHECK(nv == 3 * ntri);
            float rec[6];
            int off_sphere = 0;
            for (size_t i = 0; i < nv && fread(rec, 4, 6, f) == 6; ++i) {
                const float r = sqrtf(rec[0] * rec[0] + rec[1] * rec[1] + rec[2] * rec[2]);
                const float nr = (rec[0] * rec[3] + rec[1] * rec[4] + rec[2] * rec[5]) / r;
                if (fabsf(r - 0.9f) > 0.01f || nr < 0.95f) ++off_sphere;
            }
            CHECK(off_sphere == 0);
            fclose(f);
            remove("/tmp/roo_api_test_mesh.ply");
        }
    }

    // ---- RaycastSdf on the device vs the same march with the host containers ----
    Image<float, TargetDevice, Manage> depth(w, h), img(w, h);
    Image<float4, TargetDevice, Manage> norm(w, h);
    Mat<float,3,4> T_wc = SE3Identity();
    T_wc(2, 3) = -3.0f;
    RaycastSdf(depth, norm, img, vol, T_wc, K, 0.1f, 10.0f, 0.0f, true);
    std::vector<float> hd((size_t)w * h);
    std::vector<float4> hn((size_t)w * h);
    depth.MemcpyToHost(hd.data());
    norm.MemcpyToHost(hn.data());
    int mism = 0, nhit = 0;
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) {
            const float3 c_w = SE3Translation(T_wc);
            const float3 ray_c = K.Unproject((float)u, (float)v);
            const float3 ray_w = mulSO3(T_wc, ray_c);
            const float3 ta = div_cw(sub(hvol.bbox.Min(), c_w), ray_w), tb = div_cw(sub(hvol.bbox.Max(), c_w), ray_w);
            const float3 tmin = min3(ta, tb), tmax = max3(ta, tb);
            const float t0 = fmaxf(fmaxf(fmaxf(tmin.x, tmin.y), tmin.z), 0.1f);
            const float t1 = fminf(fminf(fminf(tmax.x, tmax.y), tmax.z), 10.0f);
            float d = 0.f;
            if (t0 < t1) {
                float lambda = t0, last = NAN, step = 0;
                const float min_step = hvol.VoxelSizeUnits().x;
                while (lambda < t1) {
                    const float sdf = hvol.GetUnitsTrilinearClamped(add(c_w, scaled(ray_w, lambda)));
                    if (sdf <= 0) {
                        if (last > 0) { lambda = lambda + step * sdf / (last - sdf); d = lambda; }
                        break;
                    }
                    step = sdf > 0 ? fmaxf(sdf, min_step) : 0.0f;
                    lambda += step;
                    last = sdf;
                }
            }
            const float got = hd[(size_t)v * w + u];
            if (d > 0) {
                ++nhit;
                const float3 g = hvol.GetUnitsBackwardDiffDxDyDz(add(c_w, scaled(ray_w, d)));
                const float len = length(g);
                const float3 n_w = len > 0 ? div_by(g, len) : make_float3(0, 0, 1);
                const float3 n_c = mulSO3inv(T_wc, n_w);
                const float4 gn = hn[(size_t)v * w + u];
                if (!(same(got, d) && gn.x == n_c.x && gn.y == n_c.y && gn.z == n_c.z && gn.w == 1.0f)) ++mism;
            } else if (!std::isnan(got)) ++mism;
        }
    CHECK(nhit > w * h / 8);
    CHECK(mism == 0);

    // ---- DepthToVbo / NormalsFromVbo vs the host formulas ----
    std::vector<float> hdep((size_t)w * h);
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) hdep[(size_t)v * w + u] = 2.0f + 0.01f * u + 0.02f * v + ((u * 7 + v * 3) % 11 == 0 ? NAN : 0.f);
    Image<float, TargetDevice, Manage> dd(w, h);
    Image<float4, TargetDevice, Manage> vbo(w, h), nrm(w, h);
    dd.MemcpyFromHost(hdep.data());
    DepthToVbo<float>(vbo, dd, K);
    NormalsFromVbo(nrm, vbo);
    std::vector<float4> hv((size_t)w * h), hnn((size_t)w * h);
    vbo.MemcpyToHost(hv.data());
    nrm.MemcpyToHost(hnn.data());
    int vb = 0;
    for (int v = 0; v < h; ++v)
        for (int u = 0; u < w; ++u) {
            const float3 P = K.Unproject((float)u, (float)v, hdep[(size_t)v * w + u]);
            const float4 g = hv[(size_t)v * w + u];
            if (!(same(g.x, P.x) && same(g.y, P.y) && same(g.z, P.z) && g.w == 1.0f)) ++vb;
        }
    CHECK(vb == 0);
    CHECK(hnn[(size_t)(h - 1) * w].w == 0.0f && hnn[w - 1].w == 0.0f && hnn[0].w == 1.0f);

    // ---- fp16 cells (config C5): SdfSphere + RaycastSdf on BoundedVolume<SDF_h> ----
    {
        CHECK(sizeof(SDF_h) == 4 && sizeof(BoundedVolume<SDF_h>) == 72);
        BoundedVolume<SDF_h, TargetDevice, Manage> volh(N, N, N, make_float3(-1, -1, -1), make_float3(1, 1, 1));
        BoundedVolume<SDF_h, TargetHost, Manage> hvolh(N, N, N, make_float3(-1, -1, -1), make_float3(1, 1, 1));
        SdfReset(volh, NAN);
        SdfSphere(volh, make_float3(0, 0, 0), 0.9f);
        CHECK(kfx_memcpy_2d(hvolh.ptr, hvolh.pitch, volh.ptr, volh.pitch, N * sizeof(SDF_h), (size_t)N * N, 2, 0) == 0);
        int badh = 0;
        for (int z = 0; z < N; ++z)
            for (int y = 0; y < N; ++y)
                for (int x = 0; x < N; ++x) {
                    const SDF_h expect(length(hvolh.VoxelPositionInUnits(x, y, z)) - 0.9f);
                    if (hvolh(x, y, z).val != expect.val || hvolh(x, y, z).w != expect.w) ++badh;
                }
        CHECK(badh == 0);
        RaycastSdf(depth, norm, img, volh, T_wc, K, 0.1f, 10.0f, 0.0f, true);
        std::vector<float> hdh((size_t)w * h);
        depth.MemcpyToHost(hdh.data());
        int nh = 0;
        for (float d : hdh) nh += std::isfinite(d) ? 1 : 0;
        CHECK(nh > w * h / 8);
        // the half volume's surface is within half precision of the fp32 one
        float worst = 0;
        for (size_t i = 0; i < hdh.size(); ++i)
            if (std::isfinite(hdh[i]) && std::isfinite(hd[i])) worst = fmaxf(worst, fabsf(hdh[i] - hd[i]));
        CHECK(worst < 5e-3f);
    }

    // ---- ROI view: SubBoundingVolume of the frustum, fuse through it ----
    const BoundingBox roi(T_wc, w, h, K, 2.2f, 3.5f);
    BoundedVolume<SDF_t> work = vol.SubBoundingVolume(roi);
    CHECK(work.w <= (size_t)N && work.pitch == vol.pitch && work.img_pitch == vol.img_pitch);
    CHECK(work.bbox.Min().z >= vol.bbox.Min().z && work.bbox.Max().z <= vol.bbox.Max().z);
    if (work.IsValid()) SdfFuse(work, dd, nrm, SE3inv(T_wc), K, 0.1f, 100.0f, 0.1f);
    CHECK(kfx_stream_synchronize(0) == 0);

    // ---- colour path: SdfReset(BoundedVolume<float>), colour SdfFuse and colour RaycastSdf ----
    {
        CHECK(sizeof(uchar3) == 3 && sizeof(Image<uchar3>) == 32 && sizeof(BoundedVolume<float>) == 72);
        BoundedVolume<SDF_t, TargetDevice, Manage> cv(N, N, N, make_float3(-1, -1, 2), make_float3(1, 1, 4));
        BoundedVolume<float, TargetDevice, Manage> colorVol(N, N, N, make_float3(-1, -1, 2), make_float3(1, 1, 4));
        BoundedVolume<float, TargetHost, Manage> hcol(N, N, N, make_float3(-1, -1, 2), make_float3(1, 1, 4));
        SdfReset(cv, NAN);
        SdfReset(colorVol);
        Image<uchar3, TargetDevice, Manage> rgb(w, h);
        std::vector<uchar3> hrgb((size_t)w * h);
        for (auto& px : hrgb) px = make_uchar3(51, 51, 51);            // grey 0.2
        rgb.MemcpyFromHost(hrgb.data());
        std::vector<float> flat((size_t)w * h, 3.0f);                   // a wall at z = 3 m
        dd.MemcpyFromHost(flat.data());
        DepthToVbo<float>(vbo, dd, K);
        NormalsFromVbo(nrm, vbo);
        const Mat<float,3,4> I = SE3Identity();
        SdfFuse(cv, colorVol, dd, nrm, I, K, rgb, I, K, 0.3f, 100.0f, 0.1f);
        CHECK(kfx_memcpy_2d(hcol.ptr, hcol.pitch, colorVol.ptr, colorVol.pitch, N * sizeof(float), (size_t)N * N, 2, 0) == 0);
        int fused = 0, wrong = 0;
        for (int z = 0; z < N; ++z)
            for (int y = 0; y < N; ++y)
                for (int x = 0; x < N; ++x) {
                    const float c = hcol(x, y, z);
                    if (c != 0.5f) { ++fused; if (std::fabs(c - 0.2f) > 1e-6f) ++wrong; }
                }
        CHECK(fused > 100 && wrong == 0);   // the 64x48 test camera sees a small patch of the 2 m volume
        RaycastSdf(depth, norm, img, cv, colorVol, I, K, 0.5f, 10.0f, 0.3f, true);
        std::vector<float> hi((size_t)w * h), hdc((size_t)w * h);
        img.MemcpyToHost(hi.data());
        depth.MemcpyToHost(hdc.data());
        int chit = 0, cbad = 0;
        for (size_t i = 0; i < hi.size(); ++i)
            if (std::isfinite(hdc[i])) { ++chit; if (std::fabs(hi[i] - 0.2f) > 1e-5f || std::fabs(hdc[i] - 3.0f) > 0.05f) ++cbad; }
        CHECK(chit > w * h / 8 && cbad == 0);
    }

    // ---- brick summary (roo::SdfSummary): tracked SdfFuse / RaycastSdf give the volume and images of the plain calls ----
    {
        const int M = 64, sw = 160, sh = 120;
        const ImageIntrinsics Ks(142.5855, 142.5855, sw / 2.0 - 0.5, sh / 2.0 - 0.5);
        const BoundingBox box(make_float3(-1, -1, 2), make_float3(1, 1, 4));
        BoundedVolume<SDF_t, TargetDevice, Manage> va(M, M, M, box), vb(M, M, M, box);
        Image<float, TargetDevice, Manage> sd(sw, sh), da(sw, sh), db(sw, sh), ia(sw, sh), ib(sw, sh);
        Image<float4, TargetDevice, Manage> sv(sw, sh), sn(sw, sh), na(sw, sh), nb(sw, sh);
        std::vector<float> wall((size_t)sw * sh, 3.5f);
        for (int v = 30; v < 90; ++v)
            for (int u = 40; u < 120; ++u) wall[(size_t)v * sw + u] = 2.8f;   // a box in front of the wall
        sd.MemcpyFromHost(wall.data());
        DepthToVbo<float>(sv, sd, Ks);
        NormalsFromVbo(sn, sv);
        const Mat<float,3,4> I = SE3Identity();
        const float tr = 2.0f * length(va.VoxelSizeUnits());
        SdfSummary summary(vb);
        SdfReset(va, NAN);
        SdfReset(vb, NAN, summary);
        for (int f = 0; f < 2; ++f) {
            SdfFuse(va, sd, sn, I, Ks, tr, 1000.0f, 0.1f);
            SdfFuse(vb, summary, sd, sn, I, Ks, tr, 1000.0f, 0.1f);
        }
        RaycastSdf(da, na, ia, va, I, Ks, 0.4f, 8.0f, tr, true);
        RaycastSdf(db, nb, ib, vb, summary, I, Ks, 0.4f, 8.0f, tr, true);
        Volume<SDF_t, TargetHost, Manage> ha(M, M, M), hb(M, M, M);
        CHECK(kfx_memcpy_2d(ha.ptr, ha.pitch, va.ptr, va.pitch, M * sizeof(SDF_t), (size_t)M * M, 2, 0) == 0);
        CHECK(kfx_memcpy_2d(hb.ptr, hb.pitch, vb.ptr, vb.pitch, M * sizeof(SDF_t), (size_t)M * M, 2, 0) == 0);
        CHECK(memcmp(ha.ptr, hb.ptr, ha.pitch * M * M) == 0);
        std::vector<float> hda((size_t)sw * sh), hdb((size_t)sw * sh);
        da.MemcpyToHost(hda.data());
        db.MemcpyToHost(hdb.data());
        int shits = 0;
        for (size_t i = 0; i < hda.size(); ++i) {
            if (std::isfinite(hda[i])) ++shits;
            CHECK((std::isnan(hda[i]) && std::isnan(hdb[i])) || hda[i] == hdb[i]);
        }
        CHECK(shits > sw * sh / 4);
    }

    // ---- kfx_frame_step: the frame's five roo:: calls as one C call (include/kfx.h), same bits, device events around its parts ----
    {
        const int M = 64, sw = 160, sh = 120;
        const ImageIntrinsics Ks(142.5855, 142.5855, sw / 2.0 - 0.5, sh / 2.0 - 0.5);
        const BoundingBox box(make_float3(-1, -1, 2), make_float3(1, 1, 4));
        BoundedVolume<SDF_t, TargetDevice, Manage> va(M, M, M, box), vb(M, M, M, box);
        Image<float, TargetDevice, Manage> raw(sw, sh), fa(sw, sh), fb(sw, sh), da(sw, sh), db(sw, sh), ia(sw, sh), ib(sw, sh);
        Image<float4, TargetDevice, Manage> va4(sw, sh), vb4(sw, sh), nna(sw, sh), nnb(sw, sh), ra(sw, sh), rb(sw, sh);
        std::vector<float> wall((size_t)sw * sh, 3.5f);
        for (int v = 30; v < 90; ++v)
            for (int u = 40; u < 120; ++u) wall[(size_t)v * sw + u] = 2.8f + 0.001f * (float)((u * 7 + v * 3) % 11);
        raw.MemcpyFromHost(wall.data());
        const float tr = 2.0f * length(va.VoxelSizeUnits());
        kfx_frame_config cfg;
        memset(&cfg, 0, sizeof(cfg));
        cfg.vol = *vb.abi(); cfg.raw = *raw.abi(); cfg.filtered = *fb.abi(); cfg.vbo = *vb4.abi(); cfg.normals = *nnb.abi();
        cfg.ray_depth = *db.abi(); cfg.ray_norm = *rb.abi(); cfg.ray_img = *ib.abi();
        cfg.K[0] = Ks.fu; cfg.K[1] = Ks.fv; cfg.K[2] = Ks.u0; cfg.K[3] = Ks.v0;
        cfg.bilateral_gs = 1.5f; cfg.bilateral_gr = 0.1f; cfg.bilateral_minval = 0.2f; cfg.bilateral_size = 3;
        cfg.near = 0.4f; cfg.far = 8.0f; cfg.trunc_dist = tr; cfg.max_w = 1000.0f; cfg.mincostheta = 0.1f;
        cfg.timing_slots = 8;
        kfx_frame* fr = nullptr;
        CHECK(kfx_frame_create(&fr, &cfg) == 0 && fr != nullptr);
        SdfReset(va, NAN);
        CHECK(kfx_frame_reset(fr, 0) == 0);
        Mat<float,3,4> T = SE3Identity();
        for (int f = 0; f < 3; ++f) {
            T(0, 3) = 0.01f * (float)f;
            BilateralFilter<float,float>(fa, raw, 1.5f, 0.1f, 3, 0.2f);
            DepthToVbo<float>(va4, fa, Ks);
            NormalsFromVbo(nna, va4);
            SdfFuse(va, fa, nna, SE3inv(T), Ks, tr, 1000.0f, 0.1f);
            RaycastSdf(da, ra, ia, va, T, Ks, 0.4f, 8.0f, tr, true);
            const Mat<float,3,4> Tinv = SE3inv(T);
            CHECK(kfx_frame_step(fr, nullptr, T.m, Tinv.m, 0, 0) == 0);
        }
        CHECK(kfx_frame_count(fr) == 3);
        float ms[3 * KFX_FRAME_TIMING_FIELDS];
        CHECK(kfx_frame_timings(fr, 0, 3, ms) == 0);
        CHECK(ms[1] > 0.f && ms[2] > 0.f && ms[3] >= ms[1] + ms[2] && ms[4] >= ms[3] * 0.999f);
        CHECK(kfx_frame_timings(fr, 0, 9, ms) == KFX_E_RANGE);
        Volume<SDF_t, TargetHost, Manage> ha(M, M, M), hb(M, M, M);
        CHECK(kfx_memcpy_2d(ha.ptr, ha.pitch, va.ptr, va.pitch, M * sizeof(SDF_t), (size_t)M * M, 2, 0) == 0);
        CHECK(kfx_memcpy_2d(hb.ptr, hb.pitch, vb.ptr, vb.pitch, M * sizeof(SDF_t), (size_t)M * M, 2, 0) == 0);
        CHECK(memcmp(ha.ptr, hb.ptr, ha.pitch * M * M) == 0);
        std::vector<float> hda((size_t)sw * sh), hdb((size_t)sw * sh);
        std::vector<float4> hna((size_t)sw * sh), hnb((size_t)sw * sh);
        da.MemcpyToHost(hda.data()); db.MemcpyToHost(hdb.data());
        nna.MemcpyToHost(hna.data()); nnb.MemcpyToHost(hnb.data());
        int fhits = 0;
        for (size_t i = 0; i < hda.size(); ++i) {
            if (std::isfinite(hda[i])) ++fhits;
            CHECK(same(hda[i], hdb[i]));
            CHECK(same(hna[i].x, hnb[i].x) && same(hna[i].y, hnb[i].y) && same(hna[i].z, hnb[i].z) && hna[i].w == hnb[i].w);
        }
        CHECK(fhits > sw * sh / 4);
        CHECK(kfx_frame_destroy(fr) == 0);
    }

    printf("roo_api_test: %s (%d ray hits checked)\n", g_fail ? "FAILED" : "all checks passed", nhit);
    return g_fail ? 1 : 0;
}
