// configs[0] "plumbing": the reference's VPL generation (rt::PhotonTracer::trace) driven by a C++
// host exactly as the reference's host would drive it, with every Scene::trace and density sample
// answered by the GPU through the C ABI (gigl_amd/host/gigl_host.hpp).
//
// usage: vpl_plumbing <in.bin> <out.bin> <fog_enabled 0|1>
// in.bin : u32 n_verts, n_tris, n_mat, n_rng, max_vpls; f32 verts[3n]; u32 tris[4m];
//          f32 materials[12k]; u8 dens[64^3]; f32 rng[n_rng]; f32 ppl_pos[3], ppl_int[3];
//          f32 maj_ext_k, abs_k, sca_k
// out.bin: u32 n_vpls, n_paths, n_rng_used, pad; VPL records (96 B each)
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../gigl_amd/host/gigl_host.hpp"

using namespace gigl_host;

template <typename T>
static void rd(FILE* f, T* p, size_t n) {
    if (n && fread(p, sizeof(T), n, f) != n) { fprintf(stderr, "short read\n"); exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "usage: %s in out fog\n", argv[0]); return 2; }
    FILE* f = fopen(argv[1], "rb");
    if (!f) { perror("in"); return 2; }
    uint32_t hd[5];
    rd(f, hd, 5);
    std::vector<float> verts(3 * (size_t)hd[0]);
    std::vector<uint32_t> tris(4 * (size_t)hd[1]);
    std::vector<PhongMaterial> mats(hd[2]);
    std::vector<uint8_t> dens(64 * 64 * 64);
    std::vector<float> rng(hd[3]);
    float ppl[6], coef[3];
    rd(f, verts.data(), verts.size());
    rd(f, tris.data(), tris.size());
    rd(f, (float*)mats.data(), 12 * (size_t)hd[2]);
    rd(f, dens.data(), dens.size());
    rd(f, rng.data(), rng.size());
    rd(f, ppl, 6);
    rd(f, coef, 3);
    fclose(f);
    const int fog_on = atoi(argv[3]);
    try {
        Scene scene;
        scene.setGeometry(verts, tris, mats, 0);
        const int res[3] = {64, 64, 64};
        scene.addFog(dens, res, coef[0], coef[1], coef[2]);
        if (!fog_on) scene.toggleFog();
        size_t pos = 0;
        bool starved = false;
        auto gen = [&]() -> float {
            if (pos < rng.size()) return rng[pos++];
            starved = true;
            return 0.0f;
        };
        // shoot_dir as DeferredRenderer::updateLights computes it (Source/Common/Renderer.cpp:222); unused by trace
        LightArray la;
        uint32_t n_paths = 0;
        rt::PhotonTracer::trace(scene, v3(ppl[0], ppl[1], ppl[2]), v3(ppl[3], ppl[4], ppl[5]), v3(0, -1, 0),
                                (int)hd[4], la, gen, &n_paths);
        if (starved) { fprintf(stderr, "rng stream exhausted\n"); return 3; }
        FILE* o = fopen(argv[2], "wb");
        if (!o) { perror("out"); return 2; }
        const uint32_t oh[4] = {(uint32_t)la.size(), n_paths, (uint32_t)pos, 0};
        fwrite(oh, 4, 4, o);
        if (la.size()) fwrite(la.pls.data(), sizeof(VPL), la.pls.size(), o);
        fclose(o);
        printf("vpl_plumbing: %d VPLs from %u paths, %zu random numbers\n", la.size(), n_paths, pos);
    } catch (const GiglError& e) {
        fprintf(stderr, "%s (code %d)\n", e.what(), e.code);
        return 1;
    }
    return 0;
}
