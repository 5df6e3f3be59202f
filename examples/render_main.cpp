// render_main.cpp -- headless analogue of the reference's main loop (ref: Source/Main.cpp:757-949) on the C ABI:
// scene set-up as Main.cpp:775-819 (with the synthetic dragon stand-in, or a glTF given on the command line),
// N x Render(), then a host framebuffer dump instead of the DX12 present (ref: Main.cpp:935-936).
//
//   g++ -std=c++17 -Iinclude -Icpugpupathtracing_amd/csrc/host examples/render_main.cpp
//       -Lcpugpupathtracing_amd/lib -lcpugpupt -Wl,-rpath,$PWD/cpugpupathtracing_amd/lib -o render_main   (one command line)
//   ./render_main [--gpus N [--collective]] [model.gltf] [width height spp [preview_every [move_at right up forward]]]
// --gpus N: ONE context over the first N GPUs of the node (cgpt_ctx_create with n_devices = N): every frame is spread over them in
// interleaved row bands and the read-back gathers the float4 bands with one RCCL exchange over xGMI; the loop below does not
// change.  --collective: take that code path with N = 1 too (what a one-GPU box can test).
// preview_every > 0 writes preview_NNNN.ppm every that many samples: the progressive display the reference gets from
// presenting data.pixels after every Render() (ref: Main.cpp:935-936, Source/DX12.cpp:277-322).
// move_at > 0 scripts the input half of Update(dt) (ref: Main.cpp:277-297, Camera::Update :104-131): after that many samples the
// camera is translated by (right, up, forward) as the A/D, Space/Shift, W/S keys would, the view changes, and the accumulator
// is reset (ref: ResetAccumulator, Main.cpp:238-243) before the remaining samples are rendered from the new position.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "cpugpupt_abi.h"
#include "gltf_loader.h"
#include "image_io.h"
#include "mesh_gen.h"
#include "scene.h"

using namespace cgpt;

#define CHECK(call)                                                                   \
    do {                                                                              \
        int rc_ = (call);                                                             \
        if (rc_ != CGPT_OK) { fprintf(stderr, "%s -> %d: %s\n", #call, rc_, cgpt_last_error(ctx)); return 1; } \
    } while (0)

int main(int argc, char** argv)
{
    int n_gpus = 1; uint32_t ctx_flags = 0;
    while (argc > 1 && std::string(argv[1]).rfind("--", 0) == 0) {
        if (std::string(argv[1]) == "--gpus" && argc > 2) { n_gpus = atoi(argv[2]); argv += 2; argc -= 2; }
        else if (std::string(argv[1]) == "--collective") { ctx_flags |= CGPT_CTX_FORCE_COLLECTIVE; argv += 1; argc -= 1; }
        else { fprintf(stderr, "unknown option %s\n", argv[1]); return 2; }
    }
    std::string model = argc > 1 && std::string(argv[1]).find(".gltf") != std::string::npos ? argv[1] : "";
    const int base = model.empty() ? 1 : 2;
    const uint32_t W = argc > base ? (uint32_t)atoi(argv[base]) : 1280, H = argc > base + 1 ? (uint32_t)atoi(argv[base + 1]) : 720;
    const uint32_t spp = argc > base + 2 ? (uint32_t)atoi(argv[base + 2]) : 64;
    const uint32_t preview_every = argc > base + 3 ? (uint32_t)atoi(argv[base + 3]) : 0;
    const uint32_t move_at = argc > base + 7 ? (uint32_t)atoi(argv[base + 4]) : 0;
    const float move_right = move_at ? (float)atof(argv[base + 5]) : 0.0f, move_up = move_at ? (float)atof(argv[base + 6]) : 0.0f;
    const float move_forward = move_at ? (float)atof(argv[base + 7]) : 0.0f;

    Mesh mesh;
    if (model.empty()) mesh = MakeDragonStandIn(6);
    else { std::string err; if (!GLTFLoader::Load(model, mesh, err)) { fprintf(stderr, "%s\n", err.c_str()); return 1; } }   // ref: Main.cpp:785
    Scene scene = MakeReferenceScene(mesh, 3, (float)W / (float)H, MeshBVH::BuildOption_SAHSplitIntervals);            // ref: Main.cpp:777-819

    cgpt_ctx* ctx = nullptr;
    // ThreadPool::Init: device_ids = NULL means devices 0 .. n_gpus-1
    if (cgpt_ctx_create(nullptr, n_gpus, ctx_flags, &ctx) != CGPT_OK) { fprintf(stderr, "%s\n", cgpt_last_error(nullptr)); return 1; }
    Scene::FlatStorage flat;
    cgpt_scene_desc desc = scene.Flatten(flat);
    CHECK(cgpt_scene_upload(ctx, &desc));

    const cgpt_settings settings = scene.AbiSettings();
    uint32_t num_accumulated = 0;                                        // data.num_accumulated
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<uint32_t> pixels((size_t)W * H);
    const uint32_t chunk = preview_every ? preview_every : 16;           // Render() calls folded into one launch
    bool moved = false;
    for (uint32_t frame = 0; frame < spp; frame += chunk) {              // the frame loop
        if (move_at && !moved && frame >= move_at) {                     // Update(dt): camera input, then ResetAccumulator() if the view changed
            moved = true;
            if (scene.camera.Move(move_right, move_up, move_forward)) {
                CHECK(cgpt_reset_accumulator(ctx));
                num_accumulated = 0;
            }
        }
        cgpt_render_params p{};
        p.width = W; p.height = H; p.row_begin = 0; p.row_end = H;
        p.first_sample = num_accumulated; p.n_samples = spp - frame < chunk ? spp - frame : chunk;
        p.seed = 0x12345678u; p.kernel = CGPT_KERNEL_AUTO;
        CHECK(cgpt_render(ctx, &scene.camera.Abi(), &settings, &p));
        num_accumulated += p.n_samples;
        if (preview_every) {                                             // DX12::CopyToBackBuffer + Present
            CHECK(cgpt_read_pixels(ctx, pixels.data(), pixels.size()));
            char name[64]; snprintf(name, sizeof(name), "preview_%04u.ppm", num_accumulated);
            std::string err; WritePPM(name, pixels.data(), W, H, err);
        }
    }
    const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    cgpt_stats st{};
    CHECK(cgpt_get_stats(ctx, &st));
    printf("%d GPU(s), %ux%u, %u spp (%u accumulated): %.1f ms/frame, %.1f Mrays/s, total energy %.3f\n", n_gpus, W, H, spp, num_accumulated, 1e3 * sec / spp,
           st.traced_rays / sec / 1e6, st.total_energy_received);

    std::vector<float> acc((size_t)W * H * 4);
    CHECK(cgpt_read_pixels(ctx, pixels.data(), pixels.size()));          // DX12::CopyToBackBuffer(data.pixels)
    CHECK(cgpt_read_accumulator(ctx, acc.data(), acc.size()));
    std::string err;
    WritePPM("render.ppm", pixels.data(), W, H, err);
    WritePFM("render.pfm", acc.data(), num_accumulated, W, H, err);
    WriteAccumulator("render.acc", acc.data(), num_accumulated, W, H, err);
    cgpt_ctx_destroy(ctx);                                               // ThreadPool::Exit
    return 0;
}
