// xpbd_headless.cpp -- headless driver: runs the solver alone (no window, no
// renderer; the reference's app/renderer side, src/app.rs + src/renderer.rs, is
// out of scope) on a seeded synthetic scene and reports body*substeps/s.
//
//   xpbd_headless --bodies 262144 --substeps 20 --frames 10 [--scene boxes|mixed|boxes-drop|mixed-drop|stacks]
//                 [--seed 1] [--mode fused|substep|contacts] [--device 0] [--dump poses.bin]
//                 [--history] [--rewind K] [--shards N]
// --shards N (with --mode contacts): the world sharded over N GPUs driven by this one process (world::ShardedWorld over
// xpbd_multi_world_*: devices --device .. --device + N - 1 over RCCL; with fewer visible devices all shards share --device and
// exchange by peer copies -- the one-GPU rehearsal).  The dump then equals the unsharded run's, byte for byte.
// --history keeps every frame on the device like the reference app's `states` vector (src/app.rs:48) and steps through
// world::Timeline; --rewind K then scrubs back to state K (0 = the initial world) before the dump.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#include "constraint_solver.hpp"

using namespace constraint_solver;

int main(int argc, char **argv)
{
    uint32_t bodies = 4096, substeps = 20, frames = 10, warmup = 2, shards = 0;
    bool history = false;
    long rewind = -1;
    uint64_t seed = 1;
    int device = 0;
    scene::Kind kind = scene::BOXES;
    uint32_t mode = XPBD_MODE_FUSED;
    std::string dump;
    for (int i = 1; i < argc; ++i) {
        auto val = [&](const char *flag) -> const char * {
            if (std::strcmp(argv[i], flag) != 0)
                return nullptr;
            if (i + 1 >= argc) {
                std::fprintf(stderr, "%s needs a value\n", flag);
                std::exit(2);
            }
            return argv[++i];
        };
        if (const char *v = val("--bodies")) bodies = (uint32_t)std::strtoul(v, nullptr, 10);
        else if (const char *v = val("--substeps")) substeps = (uint32_t)std::strtoul(v, nullptr, 10);
        else if (const char *v = val("--frames")) frames = (uint32_t)std::strtoul(v, nullptr, 10);
        else if (const char *v = val("--warmup")) warmup = (uint32_t)std::strtoul(v, nullptr, 10);
        else if (const char *v = val("--seed")) seed = std::strtoull(v, nullptr, 10);
        else if (const char *v = val("--device")) device = std::atoi(v);
        else if (const char *v = val("--scene")) kind = std::strcmp(v, "mixed") == 0 ? scene::MIXED : std::strcmp(v, "boxes-drop") == 0 ? scene::BOXES_DROP : std::strcmp(v, "mixed-drop") == 0 ? scene::MIXED_DROP : std::strcmp(v, "stacks") == 0 ? scene::BOX_STACKS : scene::BOXES;
        else if (const char *v = val("--mode")) mode = std::strcmp(v, "substep") == 0 ? XPBD_MODE_PER_SUBSTEP : std::strcmp(v, "contacts") == 0 ? XPBD_MODE_CONTACTS : XPBD_MODE_FUSED;
        else if (const char *v = val("--dump")) dump = v;
        else if (const char *v = val("--rewind")) rewind = std::strtol(v, nullptr, 10);
        else if (const char *v = val("--shards")) shards = (uint32_t)std::strtoul(v, nullptr, 10);
        else if (std::strcmp(argv[i], "--history") == 0) history = true;
        else {
            std::fprintf(stderr, "unknown argument %s\n", argv[i]);
            return 2;
        }
    }
    try {
        std::vector<rigid::Rigid> state;
        std::vector<uint32_t> shape_id;
        scene::generate(kind, seed, scene::default_grid_width(bodies), 0, bodies, state, shape_id);

        const double dt = 1.0 / 60.0; // FRAME_TIME, src/app.rs:15
        if (shards) {
            if (mode != XPBD_MODE_CONTACTS || history) {
                std::fprintf(stderr, "--shards needs --mode contacts and no --history\n");
                return 2;
            }
            world::ShardedWorld w(shards, device);
            w.set_polytopes(scene::shapes_of(kind));
            w.upload(state, shape_id);
            for (uint32_t f = 0; f < warmup; ++f)
                w.integrate(dt, substeps);
            w.synchronize();
            const auto t0 = std::chrono::steady_clock::now();
            for (uint32_t f = 0; f < frames; ++f)
                w.integrate(dt, substeps);
            w.synchronize();
            const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            w.download(state);
            double moved = 0.0;
            const auto halo = w.halo_stats(&moved);
            std::printf("{\"bodies\": %u, \"substeps\": %u, \"frames\": %u, \"mode\": \"contacts\", \"shards\": %u, \"transport\": \"%s\", "
                        "\"seconds\": %.6f, \"body_substeps_per_s\": %.4e, \"ghosts\": %llu, \"boundary\": %llu, \"plans\": %llu, "
                        "\"max_displacement\": %.4f}\n",
                        bodies, substeps, frames, shards, w.over_rccl() ? "rccl" : "local", sec, (double)bodies * substeps * frames / sec,
                        (unsigned long long)halo[2], (unsigned long long)halo[3], (unsigned long long)halo[5], moved);
            if (!dump.empty()) {
                FILE *fp = std::fopen(dump.c_str(), "wb");
                if (!fp) {
                    std::perror("dump");
                    return 1;
                }
                std::fwrite(state.data(), sizeof(rigid::Rigid), state.size(), fp);
                std::fclose(fp);
            }
            return 0;
        }
        xpbd_config cfg;
        xpbd_config_default(&cfg);
        cfg.device = device;
        cfg.mode = mode;
        world::BatchWorld w(&cfg);
        if (mode == XPBD_MODE_CONTACTS)
            w.set_polytopes(scene::shapes_of(kind)); // extension: body-body contacts
        else
            w.set_shapes(scene::shapes_of(kind));
        w.upload(state, shape_id);

        if (history)
            warmup = 0; // state 0 of the timeline is the initial world
        for (uint32_t f = 0; f < warmup; ++f)
            w.integrate(dt, substeps);
        w.synchronize();
        world::Timeline timeline(w); // pushes the current state as state 0 (only used with --history)
        const auto t0 = std::chrono::steady_clock::now();
        for (uint32_t f = 0; f < frames; ++f) {
            if (history)
                timeline.advance(dt, substeps);
            else
                w.integrate(dt, substeps);
        }
        w.synchronize();
        const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        if (history && rewind >= 0)
            timeline.seek((uint32_t)rewind);
        const double rate = (double)bodies * substeps * frames / sec;
        w.download(state);
        const auto contacts = w.contacts();
        std::printf("{\"bodies\": %u, \"substeps\": %u, \"frames\": %u, \"mode\": \"%s\", \"seconds\": %.6f, "
                    "\"body_substeps_per_s\": %.4e, \"hbm_GBps_at_412B\": %.2f, \"contacts_last_substep\": %zu}\n",
                    bodies, substeps, frames, mode == XPBD_MODE_FUSED ? "fused" : (mode == XPBD_MODE_CONTACTS ? "contacts" : "substep"), sec, rate,
                    mode == XPBD_MODE_FUSED ? 412.0 / substeps * rate / 1e9 : 412.0 * rate / 1e9, contacts.size());
        if (!dump.empty()) {
            FILE *fp = std::fopen(dump.c_str(), "wb");
            if (!fp) {
                std::perror("dump");
                return 1;
            }
            std::fwrite(state.data(), sizeof(rigid::Rigid), state.size(), fp);
            std::fclose(fp);
        }
    } catch (const Error &e) {
        std::fprintf(stderr, "xpbd error %d: %s\n", e.code, e.what());
        return 1;
    }
    return 0;
}
