// pose_shard_host.cpp -- a C++ consumer of include/gs3d_hip.h that runs the multi-GPU mode natively: one process
// per GPU, the scene loaded by rank 0 and replicated with ONE RCCL broadcast (gs_dist_broadcast_scene), camera pose
// i rendered by rank i mod world, no per-frame collective (BASELINE configs[3]).
//
//   pose_shard_host scene.ply width height poses outdir [rank world idfile [device]]
//
// Ranks find each other through `idfile`: rank 0 writes the 128-byte RCCL id there (write + rename), the others
// wait for it.  Every rank writes the poses it owns as outdir/pose_%03d.ppm (B8G8R8A8 frame -> RGB) and prints its
// frame rate; with world == 1 it is simply a batch renderer.  Poses: the default camera yawed by k * 5 degrees.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <string>
#include <thread>
#include <vector>

#include <hip/hip_runtime_api.h>

#include "gs3d_hip.h"

static void check(int rc, const char* what) {
    if (rc != GS_OK) {
        std::fprintf(stderr, "error: %s: %s\n", what, gs_last_error());
        std::exit(1);
    }
}

int main(int argc, char** argv) {
    if (argc < 6) {
        std::fprintf(stderr, "usage: %s scene.ply width height poses outdir [rank world idfile [device]]\n", argv[0]);
        return 2;
    }
    const std::string ply = argv[1], outdir = argv[5];
    const uint32_t w = static_cast<uint32_t>(std::atoi(argv[2])), h = static_cast<uint32_t>(std::atoi(argv[3]));
    const uint64_t poses = std::strtoull(argv[4], nullptr, 10);
    const int rank = argc > 6 ? std::atoi(argv[6]) : 0, world = argc > 7 ? std::atoi(argv[7]) : 1;
    const std::string idfile = argc > 8 ? argv[8] : "";
    const int device = argc > 9 ? std::atoi(argv[9]) : 0;

    uint8_t id[GS_DIST_ID_BYTES];
    if (rank == 0) {
        check(gs_dist_unique_id(id), "gs_dist_unique_id");
        if (!idfile.empty()) {
            std::ofstream(idfile + ".tmp", std::ios::binary).write(reinterpret_cast<const char*>(id), sizeof id);
            std::rename((idfile + ".tmp").c_str(), idfile.c_str());
        }
    } else {
        for (int tries = 0;; ++tries) {
            std::ifstream f(idfile, std::ios::binary);
            if (f && f.read(reinterpret_cast<char*>(id), sizeof id)) break;
            if (tries > 600) {
                std::fprintf(stderr, "error: rank %d never saw %s\n", rank, idfile.c_str());
                return 1;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(100));
        }
    }
    gs_dist* dist = nullptr;
    check(gs_dist_create(id, rank, world, device, &dist), "gs_dist_create");

    gs_scene* loaded = nullptr;
    if (rank == 0) check(gs_scene_load_ply(ply.c_str(), device, &loaded), "gs_scene_load_ply");
    gs_scene* scene = nullptr;
    check(gs_dist_broadcast_scene(dist, loaded, 0, &scene), "gs_dist_broadcast_scene");
    // what the collective saw: every rank took part, every rank holds the root's scene (one JSON line per rank)
    gs_dist_report rep{};
    check(gs_dist_verify(dist, scene, &rep), "gs_dist_verify");
    std::printf("{\"rccl\": {\"rank\": %d, \"ranks\": %llu, \"world_size\": %llu, \"version\": %u, \"blob_MB\": %.1f, \"broadcast_ms\": %.3f, "
                "\"blob_checksums_equal\": %s, \"blob_checksum\": %llu}}\n",
                rank, static_cast<unsigned long long>(rep.ranks), static_cast<unsigned long long>(rep.world), rep.rccl_version,
                rep.broadcast_bytes / 1e6, rep.broadcast_ms, rep.checksums_equal ? "true" : "false",
                static_cast<unsigned long long>(rep.checksum));
    if (rep.ranks != static_cast<uint64_t>(world) || !rep.checksums_equal) {
        std::fprintf(stderr, "error: the collective saw %llu of %d ranks, replicas %s\n", static_cast<unsigned long long>(rep.ranks), world,
                     rep.checksums_equal ? "equal" : "DIFFER");
        return 1;
    }

    gs_renderer* rend = nullptr;
    check(gs_renderer_create(scene, &rend), "gs_renderer_create");
    void* d_bgra = nullptr;
    if (hipMalloc(&d_bgra, static_cast<size_t>(w) * h * 4) != hipSuccess) return 1;
    std::vector<uint8_t> bgra(static_cast<size_t>(w) * h * 4), rgb(static_cast<size_t>(w) * h * 3);

    const auto t0 = std::chrono::steady_clock::now();
    uint64_t mine = 0;
    for (uint64_t k = static_cast<uint64_t>(rank); k < poses; k += static_cast<uint64_t>(world), ++mine) {
        gs_camera cam{};
        // operation for operation what dist.py::pose_quaternion does (math.radians(x) = x * (pi / 180)), so that both
        // hosts hand the renderer the same float quaternion and the frames can be compared exactly
        const double a = (5.0 * static_cast<double>(k)) * (3.14159265358979323846 / 180.0) / 2.0;
        cam.rotation[0] = static_cast<float>(std::cos(a));  // (w, x, y, z): yaw about world y
        cam.rotation[2] = static_cast<float>(std::sin(a));
        cam.fov = 45.0f;
        cam.near_plane = 0.1f;
        cam.far_plane = 1000.0f;
        gs_uniforms u{};
        check(gs_camera_uniforms(&cam, w, h, &u), "gs_camera_uniforms");
        check(gs_render(rend, &u, nullptr, static_cast<uint8_t*>(d_bgra)), "gs_render");
        check(gs_synchronize(rend), "gs_synchronize");
        if (hipMemcpy(bgra.data(), d_bgra, bgra.size(), hipMemcpyDeviceToHost) != hipSuccess) return 1;
        for (size_t i = 0; i < static_cast<size_t>(w) * h; ++i) {
            rgb[3 * i + 0] = bgra[4 * i + 2];
            rgb[3 * i + 1] = bgra[4 * i + 1];
            rgb[3 * i + 2] = bgra[4 * i + 0];
        }
        char path[4096];
        std::snprintf(path, sizeof path, "%s/pose_%03llu.ppm", outdir.c_str(), static_cast<unsigned long long>(k));
        std::ofstream f(path, std::ios::binary);
        f << "P6\n" << w << " " << h << "\n255\n";
        f.write(reinterpret_cast<const char*>(rgb.data()), static_cast<std::streamsize>(rgb.size()));
    }
    const double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("rank %d/%d: %llu of %llu poses (gs_dist_pose_count %llu), %llu Gaussians, %.1f frames/s incl. readback\n", rank,
                world, static_cast<unsigned long long>(mine), static_cast<unsigned long long>(poses),
                static_cast<unsigned long long>(gs_dist_pose_count(dist, poses)),
                static_cast<unsigned long long>(gs_scene_num_vertices(scene)), mine / s);
    (void)hipFree(d_bgra);
    gs_renderer_destroy(rend);
    if (scene != loaded) gs_scene_destroy(scene);
    gs_scene_destroy(loaded);
    gs_dist_destroy(dist);
    return 0;
}
