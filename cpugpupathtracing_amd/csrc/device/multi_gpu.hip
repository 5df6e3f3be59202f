// multi_gpu.hip -- one host process, up to 8 MI355X: the multi-device context behind cgpt_ctx_create(device_ids, n_devices > 1).
//
// The reference is ONE C++ process whose main loop calls Render() and then hands data.pixels to the presenter
// (ref: Source/Main.cpp:773 ThreadPool::Init, :753-754 Dispatch / WaitAll, :834-836, :935 CopyToBackBuffer).  A maintainer who
// follows INTEGRATION.md keeps that shape: one context, one cgpt_render per frame, one cgpt_read_pixels -- and the context
// spreads the frame over the GPUs of the node:
//   * the image is cut into bands of `band_rows` rows dealt round-robin over the devices (cgpt_render_params.interleave_*:
//     every GPU gets the same mix of sky, mesh and ground rows); the scene is replicated by cgpt_scene_upload;
//   * cgpt_render enqueues every device's kernels without waiting in between, then waits for all of them;
//   * cgpt_read_accumulator / cgpt_read_pixels run ONE grouped RCCL exchange -- every device ncclSend()s its float4 band,
//     device 0 ncclRecv()s them (a gather with per-rank counts; the bands differ by up to band_rows rows) -- over xGMI, then a
//     row-reorder kernel on device 0 puts the bands back into image order.  Nothing is exchanged during tracing.
// RNG streams are keyed by the global pixel index, so the gathered image is bit-identical to a one-GPU render.
// bench.py --gpus N (launched as a plain command) runs through this context; under torch.distributed.run it keeps one process per
// GPU instead (cpugpupathtracing_amd/distributed.py).
// Host threads: the context owns one persistent worker per device beyond the first (started by cgpt_ctx_create, joined by
// cgpt_ctx_destroy); cgpt_render hands every worker its device's enqueue and does device 0's itself, so a one-sample-per-call
// host (the reference's own main loop, ref: Main.cpp:825-942) pays a condition-variable wake per frame, not a thread spawn.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <cstring>
#include <exception>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "cpugpupt_abi.h"
#include "ctx_internal.h"

namespace cgpt {

static constexpr uint32_t kMaxRanks = 8;

// Persistent per-device host workers: rank r >= 1 runs `job(arg, r)` on its own thread, the caller runs rank 0's share.
struct WorkerPool {
    std::vector<std::thread> threads;
    std::mutex m;
    std::condition_variable cv_job, cv_done;
    uint64_t generation = 0;
    uint32_t pending = 0;
    bool exit = false;
    void (*job)(void*, uint32_t) = nullptr;
    void* arg = nullptr;

    void Start(uint32_t n_ranks)                                              // may throw (std::system_error): only called under cgpt_ctx_create's try
    {
        for (uint32_t r = 1; r < n_ranks; ++r) threads.emplace_back([this, r]() { Loop(r); });
    }
    void Loop(uint32_t r)
    {
        uint64_t seen = 0;
        for (;;) {
            void (*fn)(void*, uint32_t); void* a;
            {
                std::unique_lock<std::mutex> lk(m);
                cv_job.wait(lk, [&]() { return exit || generation != seen; });
                if (exit) return;
                seen = generation; fn = job; a = arg;
            }
            fn(a, r);                                                         // jobs do not throw (they catch and record a status)
            {
                std::lock_guard<std::mutex> lk(m);
                if (--pending == 0) cv_done.notify_one();
            }
        }
    }
    void Run(void (*fn)(void*, uint32_t), void* a)                            // all ranks; returns when every rank's job has returned
    {
        const uint32_t n_workers = (uint32_t)threads.size();
        if (n_workers) {
            std::lock_guard<std::mutex> lk(m);
            job = fn; arg = a; pending = n_workers; ++generation;
        }
        if (n_workers) cv_job.notify_all();
        fn(a, 0);
        if (n_workers) {
            std::unique_lock<std::mutex> lk(m);
            cv_done.wait(lk, [&]() { return pending == 0; });
        }
    }
    void Stop()
    {
        { std::lock_guard<std::mutex> lk(m); exit = true; }
        cv_job.notify_all();
        for (std::thread& t : threads) if (t.joinable()) t.join();
        threads.clear();
    }
};

struct DeviceGroup {
    std::vector<cgpt_ctx*> members;        // one-device contexts, rank order
    std::vector<ncclComm_t> comms;         // empty with CGPT_CTX_GATHER_PEER_COPY
    WorkerPool workers;
    bool peer_copy = false;
    uint32_t band_rows_cfg = 4;            // configured band height (cgpt_set_tuning "band_rows")
    uint32_t band_rows = 4;                // effective band height of the current frame: the configured one, halved until every device owns rows
    // framebuffer of the last render
    uint32_t width = 0, height = 0;
    std::vector<uint32_t> n_rows;          // rows of each member's band
    uint32_t num_accumulated = 0;
    uint32_t last_debug_mode = 0;
    uint32_t last_kernel = 0;
    // device 0: staging (all bands, rank after rank) and the gathered full frame
    float4* d_staging = nullptr; float4* d_full = nullptr; uint32_t* d_pix_staging = nullptr; uint32_t* d_full_pixels = nullptr;
    uint32_t* d_rank_base = nullptr;       // first staging row of every rank (8 words)
    size_t alloc_pixels = 0;
    bool gathered = false, pixels_valid = false;
    uint32_t gathers = 0;
    double gather_ms = 0.0;                // summed duration of the exchanges (hipEvents on device 0's stream)
    hipEvent_t ev_gather0 = nullptr, ev_gather1 = nullptr;
};

namespace {

#define G_HIP(ctx, expr)                                                                                        \
    do {                                                                                                        \
        hipError_t e_ = (expr);                                                                                 \
        if (e_ != hipSuccess) return CtxFail((ctx), CGPT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

// band-ordered rows of all ranks (rank after rank, each in its own compact order) -> image order
__global__ void __launch_bounds__(256) reorder_rows_f4(const float4* __restrict__ staging, float4* __restrict__ full, uint32_t width, uint32_t height,
                                                        uint32_t band_rows, uint32_t n_ranks, const uint32_t* __restrict__ rank_base_rows)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t band = y / band_rows, r = band % n_ranks;
    const uint32_t local_row = (band / n_ranks) * band_rows + y % band_rows;   // same enumeration as cgpt_render_params.interleave_*
    full[i] = staging[((size_t)rank_base_rows[r] + local_row) * width + x];
}
__global__ void __launch_bounds__(256) reorder_rows_u32(const uint32_t* __restrict__ staging, uint32_t* __restrict__ full, uint32_t width, uint32_t height,
                                                         uint32_t band_rows, uint32_t n_ranks, const uint32_t* __restrict__ rank_base_rows)
{
    const size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
    if (i >= (size_t)width * height) return;
    const uint32_t y = (uint32_t)(i / width), x = (uint32_t)(i - (size_t)y * width);
    const uint32_t band = y / band_rows, r = band % n_ranks;
    const uint32_t local_row = (band / n_ranks) * band_rows + y % band_rows;
    full[i] = staging[((size_t)rank_base_rows[r] + local_row) * width + x];
}

uint32_t RowsOfRank(uint32_t height, uint32_t band_rows, uint32_t n, uint32_t r)
{
    uint32_t rows = 0;
    for (uint64_t first = (uint64_t)r * band_rows; first < height; first += (uint64_t)n * band_rows) rows += std::min<uint32_t>(band_rows, height - (uint32_t)first);
    return rows;
}

// band height of a frame: the configured one, halved until every device owns rows
uint32_t EffectiveBandRows(uint32_t cfg, uint32_t n, uint32_t height)
{
    uint32_t band_rows = cfg;
    while (band_rows > 1 && (uint64_t)n * band_rows > height) band_rows /= 2;
    return band_rows;
}

// (re)derives the tiling of a width x height frame from the configured band height
void SetTiling(DeviceGroup* g, uint32_t width, uint32_t height)
{
    const uint32_t n = (uint32_t)g->members.size();
    g->band_rows = EffectiveBandRows(g->band_rows_cfg, n, height);
    g->width = width; g->height = height;
    for (uint32_t r = 0; r < n; ++r) g->n_rows[r] = RowsOfRank(height, g->band_rows, n, r);
}

void FreeGathered(DeviceGroup* g)
{
    (void)hipFree(g->d_staging); (void)hipFree(g->d_full); (void)hipFree(g->d_pix_staging); (void)hipFree(g->d_full_pixels); (void)hipFree(g->d_rank_base);
    g->d_staging = g->d_full = nullptr; g->d_pix_staging = g->d_full_pixels = nullptr; g->d_rank_base = nullptr; g->alloc_pixels = 0;
}

// copies the members' message (if any) into the group context and returns rc
int Propagate(cgpt_ctx* ctx, cgpt_ctx* member, int rc)
{
    if (rc != CGPT_OK) ctx->error = std::string("device ") + std::to_string(member->device) + ": " + member->error;
    return rc;
}

// The one collective: every rank's band to device 0, then image order.  `pixels`: gather the RGBA8 band as well (debug views,
// where data.pixels is not a function of the accumulator).
int Gather(cgpt_ctx* ctx, bool pixels)
{
    DeviceGroup* g = ctx->group;
    const uint32_t n = (uint32_t)g->members.size();
    if (g->width == 0) return CtxFail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    if (g->gathered && (!pixels || g->pixels_valid)) return CGPT_OK;
    cgpt_ctx* root = g->members[0];
    const size_t n_px = (size_t)g->width * g->height;
    G_HIP(ctx, hipSetDevice(root->device));
    if (g->alloc_pixels < n_px) {
        FreeGathered(g);
        G_HIP(ctx, hipMalloc((void**)&g->d_staging, n_px * sizeof(float4)));
        G_HIP(ctx, hipMalloc((void**)&g->d_full, n_px * sizeof(float4)));
        G_HIP(ctx, hipMalloc((void**)&g->d_pix_staging, n_px * sizeof(uint32_t)));
        G_HIP(ctx, hipMalloc((void**)&g->d_full_pixels, n_px * sizeof(uint32_t)));
        G_HIP(ctx, hipMalloc((void**)&g->d_rank_base, kMaxRanks * sizeof(uint32_t)));
        g->alloc_pixels = n_px;
    }
    uint32_t base[kMaxRanks] = { 0 };
    for (uint32_t r = 1; r < n; ++r) base[r] = base[r - 1] + g->n_rows[r - 1];

    G_HIP(ctx, hipEventRecord(g->ev_gather0, root->stream));
    if (g->peer_copy) {
        for (uint32_t r = 0; r < n; ++r) {
            cgpt_ctx* m = g->members[r];
            const size_t count = (size_t)g->n_rows[r] * g->width;
            G_HIP(ctx, hipMemcpyPeerAsync(g->d_staging + (size_t)base[r] * g->width, root->device, m->d_accumulator, m->device, count * sizeof(float4), root->stream));
            if (pixels) G_HIP(ctx, hipMemcpyPeerAsync(g->d_pix_staging + (size_t)base[r] * g->width, root->device, m->d_pixels, m->device, count * sizeof(uint32_t), root->stream));
        }
    } else {
        // one grouped exchange: rank r sends on its own stream, rank 0 receives all of them (its own band included) on its stream.
        // A group that was opened is always closed: the first failure is kept and reported after ncclGroupEnd.
        ncclResult_t first = ncclGroupStart();
        const char* what = "ncclGroupStart";
        if (first == ncclSuccess) {
            auto keep = [&](ncclResult_t r, const char* w) { if (first == ncclSuccess && r != ncclSuccess) { first = r; what = w; } };
            for (uint32_t r = 0; r < n && first == ncclSuccess; ++r) {
                cgpt_ctx* m = g->members[r];
                const size_t count = (size_t)g->n_rows[r] * g->width;
                keep(ncclSend(m->d_accumulator, count * 4, ncclFloat, 0, g->comms[r], m->stream), "ncclSend");
                keep(ncclRecv(g->d_staging + (size_t)base[r] * g->width, count * 4, ncclFloat, (int)r, g->comms[0], root->stream), "ncclRecv");
                if (pixels) {
                    keep(ncclSend(m->d_pixels, count, ncclUint32, 0, g->comms[r], m->stream), "ncclSend");
                    keep(ncclRecv(g->d_pix_staging + (size_t)base[r] * g->width, count, ncclUint32, (int)r, g->comms[0], root->stream), "ncclRecv");
                }
            }
            keep(ncclGroupEnd(), "ncclGroupEnd");
        }
        if (first != ncclSuccess) return CtxFail(ctx, CGPT_ERR_HIP, "%s failed: %s", what, ncclGetErrorString(first));
    }
    G_HIP(ctx, hipSetDevice(root->device));
    uint32_t* const d_base = g->d_rank_base;
    G_HIP(ctx, hipMemcpyAsync(d_base, base, n * sizeof(uint32_t), hipMemcpyHostToDevice, root->stream));
    const dim3 grid((uint32_t)((n_px + 255u) / 256u)), block(256);
    hipLaunchKernelGGL(reorder_rows_f4, grid, block, 0, root->stream, (const float4*)g->d_staging, g->d_full, g->width, g->height, g->band_rows, n, (const uint32_t*)d_base);
    if (pixels) hipLaunchKernelGGL(reorder_rows_u32, grid, block, 0, root->stream, (const uint32_t*)g->d_pix_staging, g->d_full_pixels, g->width, g->height, g->band_rows, n, (const uint32_t*)d_base);
    G_HIP(ctx, hipGetLastError());
    G_HIP(ctx, hipEventRecord(g->ev_gather1, root->stream));
    for (uint32_t r = 0; r < n; ++r) {                                         // the senders' streams too: their bands are free again
        G_HIP(ctx, hipSetDevice(g->members[r]->device));
        G_HIP(ctx, hipStreamSynchronize(g->members[r]->stream));
    }
    G_HIP(ctx, hipSetDevice(root->device));
    G_HIP(ctx, hipStreamSynchronize(root->stream));
    float ms = 0.0f;
    if (hipEventElapsedTime(&ms, g->ev_gather0, g->ev_gather1) == hipSuccess) g->gather_ms += ms;
    g->gathered = true; g->pixels_valid = pixels;
    g->gathers++;
    return CGPT_OK;
}

struct EnqueueJob {
    DeviceGroup* g;
    const cgpt_camera* camera; const cgpt_settings* settings; const cgpt_render_params* p;
    int rcs[kMaxRanks];
};
void EnqueueRank(void* arg, uint32_t r)                                       // every thread touches its own member context only
{
    EnqueueJob* j = static_cast<EnqueueJob*>(arg);
    cgpt_ctx* m = j->g->members[r];
    try {
        cgpt_render_params q = *j->p;
        q.interleave_rows = j->g->band_rows; q.interleave_count = (uint32_t)j->g->members.size(); q.interleave_index = r;
        j->rcs[r] = RenderEnqueue(m, j->camera, j->settings, &q);
    } catch (const std::exception& e) {
        j->rcs[r] = CtxFail(m, CGPT_ERR_INVALID, "render enqueue: %s", e.what());
    } catch (...) {
        j->rcs[r] = CtxFail(m, CGPT_ERR_INVALID, "render enqueue: unknown exception");
    }
}

}  // namespace

cgpt_ctx* GroupFirstMember(cgpt_ctx* ctx) { return ctx->group->members[0]; }
cgpt_ctx* GroupFirstMemberOrNull(cgpt_ctx* ctx) { return ctx && ctx->group ? ctx->group->members[0] : nullptr; }
int GroupForwarded(cgpt_ctx* ctx, int rc) { return Propagate(ctx, ctx->group->members[0], rc); }

int GroupCreate(const int* device_ids, int n_devices, uint32_t flags, cgpt_ctx** out)
{
    cgpt_ctx* ctx = new (std::nothrow) cgpt_ctx;
    DeviceGroup* g = new (std::nothrow) DeviceGroup;
    if (!ctx || !g) { delete ctx; delete g; return CreateFail(CGPT_ERR_INVALID, "out of host memory"); }
    ctx->group = g;
    g->peer_copy = (flags & CGPT_CTX_GATHER_PEER_COPY) != 0;
    auto fail = [&](int rc) { GroupDestroy(ctx); delete ctx; return rc; };
    try {
        std::vector<int> devs(n_devices);
        for (int i = 0; i < n_devices; ++i) devs[i] = device_ids ? device_ids[i] : i;
        if (!g->peer_copy)                                                      // RCCL wants one rank per GPU
            for (int i = 0; i < n_devices; ++i)
                for (int j = 0; j < i; ++j)
                    if (devs[i] == devs[j])
                        return fail(CreateFail(CGPT_ERR_INVALID, "device %d listed twice (only a CGPT_CTX_GATHER_PEER_COPY context may share a device between ranks)", devs[i]));
        g->members.reserve(n_devices);
        g->n_rows.assign(n_devices, 0);
        for (int i = 0; i < n_devices; ++i) {
            cgpt_ctx* m = nullptr;
            const int rc = cgpt_ctx_create(&devs[i], 1, 0, &m);                 // its failure text is the create error already
            if (rc != CGPT_OK) return fail(rc);
            g->members.push_back(m);
        }
        hipError_t e = hipSetDevice(g->members[0]->device);
        if (e == hipSuccess) e = hipEventCreate(&g->ev_gather0);
        if (e == hipSuccess) e = hipEventCreate(&g->ev_gather1);
        if (e != hipSuccess) return fail(CreateFail(CGPT_ERR_HIP, "multi-device context events: %s", hipGetErrorString(e)));
        if (!g->peer_copy) {
            g->comms.assign(n_devices, nullptr);
            const ncclResult_t r = ncclCommInitAll(g->comms.data(), n_devices, devs.data());
            if (r != ncclSuccess) {
                g->comms.clear();
                return fail(CreateFail(CGPT_ERR_HIP, "ncclCommInitAll over %d devices failed: %s", n_devices, ncclGetErrorString(r)));
            }
        }
        g->workers.Start((uint32_t)n_devices);
    } catch (const std::exception& e) {
        return fail(CreateFail(CGPT_ERR_INVALID, "cgpt_ctx_create: %s", e.what()));
    }
    *out = ctx;
    return CGPT_OK;
}

void GroupDestroy(cgpt_ctx* ctx)
{
    DeviceGroup* g = ctx->group;
    if (!g) return;
    g->workers.Stop();
    for (cgpt_ctx* m : g->members) { (void)hipSetDevice(m->device); (void)hipStreamSynchronize(m->stream); }
    for (ncclComm_t c : g->comms) if (c) (void)ncclCommDestroy(c);
    if (!g->members.empty()) {
        (void)hipSetDevice(g->members[0]->device); FreeGathered(g);
        if (g->ev_gather0) (void)hipEventDestroy(g->ev_gather0);
        if (g->ev_gather1) (void)hipEventDestroy(g->ev_gather1);
    }
    for (cgpt_ctx* m : g->members) (void)cgpt_ctx_destroy(m);
    delete g;
    ctx->group = nullptr;
}

int GroupSceneUpload(cgpt_ctx* ctx, const cgpt_scene_desc* scene)
{
    for (cgpt_ctx* m : ctx->group->members) { const int rc = cgpt_scene_upload(m, scene); if (rc != CGPT_OK) return Propagate(ctx, m, rc); }
    ctx->has_scene = true;
    return CGPT_OK;
}

int GroupUpdateMaterials(cgpt_ctx* ctx, const cgpt_material* materials, uint32_t n)
{
    for (cgpt_ctx* m : ctx->group->members) { const int rc = cgpt_scene_update_materials(m, materials, n); if (rc != CGPT_OK) return Propagate(ctx, m, rc); }
    return CGPT_OK;
}

int GroupRender(cgpt_ctx* ctx, const cgpt_camera* camera, const cgpt_settings* settings, const cgpt_render_params* p)
{
    DeviceGroup* g = ctx->group;
    if (!camera || !settings || !p) return CtxFail(ctx, CGPT_ERR_INVALID, "null argument");
    if (p->row_begin != 0 || p->row_end != p->height || p->interleave_rows || p->interleave_count || p->interleave_index)
        return CtxFail(ctx, CGPT_ERR_INVALID, "a multi-device context tiles the image itself: pass row_begin = 0, row_end = height and no interleave");
    const uint32_t n = (uint32_t)g->members.size();
    if (p->height < n) return CtxFail(ctx, CGPT_ERR_INVALID, "image of %u rows is too small for %u devices", p->height, n);
    if (g->width != p->width || g->height != p->height || g->band_rows != EffectiveBandRows(g->band_rows_cfg, n, p->height)) SetTiling(g, p->width, p->height);
    g->gathered = false; g->pixels_valid = false;
    // Enqueue everywhere, then wait everywhere: the devices render side by side.  One host thread per device does the enqueueing
    // (the wavefront pipeline is ~60 launches per batch: enqueued one device after the other, the eighth GPU would start several
    // milliseconds after the first).
    EnqueueJob job{ g, camera, settings, p, { 0 } };
    g->workers.Run(EnqueueRank, &job);
    int first_error = CGPT_OK;
    for (uint32_t r = 0; r < n; ++r)
        if (job.rcs[r] != CGPT_OK && first_error == CGPT_OK) first_error = Propagate(ctx, g->members[r], job.rcs[r]);
    for (uint32_t r = 0; r < n; ++r) {                                         // also after a failed enqueue elsewhere: nothing stays pending
        const int rc = RenderFinish(g->members[r]);
        if (rc != CGPT_OK && first_error == CGPT_OK) first_error = Propagate(ctx, g->members[r], rc);
    }
    if (first_error != CGPT_OK) return first_error;
    g->num_accumulated = p->n_samples ? p->first_sample + p->n_samples : g->num_accumulated;
    g->last_debug_mode = settings->debug_render_mode;
    g->last_kernel = g->members[0]->last_kernel;
    return CGPT_OK;
}

int GroupResetAccumulator(cgpt_ctx* ctx)
{
    for (cgpt_ctx* m : ctx->group->members) { const int rc = cgpt_reset_accumulator(m); if (rc != CGPT_OK) return Propagate(ctx, m, rc); }
    ctx->group->num_accumulated = 0; ctx->group->gathered = false;
    return CGPT_OK;
}

int GroupReadAccumulator(cgpt_ctx* ctx, float* dst, size_t n_floats)
{
    DeviceGroup* g = ctx->group;
    const size_t n = (size_t)g->width * g->height * 4;
    if (g->width == 0) return CtxFail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    if (!dst || n_floats != n) return CtxFail(ctx, CGPT_ERR_INVALID, "expected a buffer of %zu floats", n);
    const int rc = Gather(ctx, false);
    if (rc != CGPT_OK) return rc;
    G_HIP(ctx, hipSetDevice(g->members[0]->device));
    G_HIP(ctx, hipMemcpy(dst, g->d_full, n * sizeof(float), hipMemcpyDeviceToHost));
    return CGPT_OK;
}

int GroupReadPixels(cgpt_ctx* ctx, uint32_t* dst, size_t n_pixels)
{
    DeviceGroup* g = ctx->group;
    const size_t n = (size_t)g->width * g->height;
    if (g->width == 0) return CtxFail(ctx, CGPT_ERR_INVALID, "nothing rendered yet");
    if (!dst || n_pixels != n) return CtxFail(ctx, CGPT_ERR_INVALID, "expected a buffer of %zu pixels", n);
    // data.pixels = Vec4ToUint(accumulator / num_accumulated) (ref: Main.cpp:741): packed on device 0 from the gathered sums, so the
    // float4 framebuffer is the only thing that crosses xGMI; the debug views' pixels are not a function of the sums and are gathered
    const bool debug = g->last_debug_mode != 0u;
    const int rc = Gather(ctx, debug);
    if (rc != CGPT_OK) return rc;
    cgpt_ctx* root = g->members[0];
    G_HIP(ctx, hipSetDevice(root->device));
    if (!debug) {
        G_HIP(ctx, LaunchPackPixels(g->d_full, g->d_full_pixels, n, g->num_accumulated, root->stream));
        G_HIP(ctx, hipStreamSynchronize(root->stream));
    }
    G_HIP(ctx, hipMemcpy(dst, g->d_full_pixels, n * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return CGPT_OK;
}

int GroupWriteAccumulator(cgpt_ctx* ctx, const cgpt_render_params* p, const float* src, size_t n_floats, uint32_t num_accumulated)
{
    DeviceGroup* g = ctx->group;
    if (!p || !src) return CtxFail(ctx, CGPT_ERR_INVALID, "null argument");
    if (p->row_begin != 0 || p->row_end != p->height || p->interleave_rows || p->interleave_count || p->interleave_index || p->width == 0 || p->height == 0)
        return CtxFail(ctx, CGPT_ERR_INVALID, "a multi-device context restores whole frames: pass row_begin = 0, row_end = height and no interleave");
    if (n_floats != (size_t)p->width * p->height * 4) return CtxFail(ctx, CGPT_ERR_INVALID, "expected %zu floats", (size_t)p->width * p->height * 4);
    const uint32_t n = (uint32_t)g->members.size();
    if (p->height < n) return CtxFail(ctx, CGPT_ERR_INVALID, "image of %u rows is too small for %u devices", p->height, n);
    SetTiling(g, p->width, p->height);
    const uint32_t band_rows = g->band_rows;
    std::vector<float> band;
    for (uint32_t r = 0; r < n; ++r) {
        band.resize((size_t)g->n_rows[r] * p->width * 4);
        size_t out_row = 0;
        for (uint64_t first = (uint64_t)r * band_rows; first < p->height; first += (uint64_t)n * band_rows)
            for (uint32_t j = 0; j < band_rows && first + j < p->height; ++j, ++out_row)
                memcpy(band.data() + out_row * p->width * 4, src + (first + j) * (size_t)p->width * 4, (size_t)p->width * 16);
        cgpt_render_params q = *p;
        q.interleave_rows = band_rows; q.interleave_count = n; q.interleave_index = r;
        const int rc = cgpt_write_accumulator(g->members[r], &q, band.data(), band.size(), num_accumulated);
        if (rc != CGPT_OK) return Propagate(ctx, g->members[r], rc);
    }
    g->num_accumulated = num_accumulated; g->last_debug_mode = 0; g->gathered = false; g->pixels_valid = false;
    return CGPT_OK;
}

int GroupDevicePtr(cgpt_ctx* ctx, bool pixels, void** ptr, size_t* n_bytes)
{
    DeviceGroup* g = ctx->group;
    int rc = Gather(ctx, pixels && g->last_debug_mode != 0u);
    if (rc != CGPT_OK) return rc;
    const size_t n = (size_t)g->width * g->height;
    if (pixels && g->last_debug_mode == 0u) {
        cgpt_ctx* root = g->members[0];
        G_HIP(ctx, hipSetDevice(root->device));
        G_HIP(ctx, LaunchPackPixels(g->d_full, g->d_full_pixels, n, g->num_accumulated, root->stream));
        G_HIP(ctx, hipStreamSynchronize(root->stream));
    }
    *ptr = pixels ? (void*)g->d_full_pixels : (void*)g->d_full;               // on device_ids[0]
    *n_bytes = n * (pixels ? sizeof(uint32_t) : sizeof(float4));
    return CGPT_OK;
}

int GroupGetStats(cgpt_ctx* ctx, cgpt_stats* out)
{
    DeviceGroup* g = ctx->group;
    memset(out, 0, sizeof(*out));
    uint32_t r = 0;
    for (cgpt_ctx* m : g->members) {
        cgpt_stats s;
        const int rc = cgpt_get_stats(m, &s);
        if (rc != CGPT_OK) return Propagate(ctx, m, rc);
        out->traced_rays += s.traced_rays; out->inner_steps += s.inner_steps; out->tri_tests += s.tri_tests;
        out->bvh_depth_sum += s.bvh_depth_sum; out->closest_hits += s.closest_hits; out->total_energy_received += s.total_energy_received;
        out->kernel_launches += s.kernel_launches;
        out->kernel_ms = std::max(out->kernel_ms, s.kernel_ms);                // the devices run side by side: the frame takes as long as the slowest
        out->dominant_launches += s.dominant_launches; out->dominant_ms += s.dominant_ms;
        out->dominant_waves_per_simd = s.dominant_waves_per_simd;
        out->dominant_round0_ms += s.dominant_round0_ms; out->dominant_round0_launches += s.dominant_round0_launches;
        out->device_ms[r++] = s.kernel_ms;
    }
    out->num_accumulated = g->num_accumulated;
    out->gather_ms = g->gather_ms; out->gathers = g->gathers;
    out->n_devices = (uint32_t)g->members.size(); out->rccl_ranks = (uint32_t)g->comms.size();
    out->last_kernel = g->last_kernel;
    return CGPT_OK;
}

int GroupResetStats(cgpt_ctx* ctx)
{
    for (cgpt_ctx* m : ctx->group->members) { const int rc = cgpt_reset_stats(m); if (rc != CGPT_OK) return Propagate(ctx, m, rc); }
    ctx->group->gather_ms = 0.0; ctx->group->gathers = 0;
    return CGPT_OK;
}

int GroupSetTuning(cgpt_ctx* ctx, const char* name, uint32_t value)
{
    DeviceGroup* g = ctx->group;
    if (name && strcmp(name, "band_rows") == 0) {
        if (value == 0 || value > 1024) return CtxFail(ctx, CGPT_ERR_INVALID, "band_rows %u outside [1, 1024]", value);
        if (value == g->band_rows_cfg) return CGPT_OK;
        const uint32_t n = (uint32_t)g->members.size();
        if (g->width == 0 || EffectiveBandRows(value, n, g->height) == g->band_rows) { g->band_rows_cfg = value; return CGPT_OK; }   // nothing to move
        // A frame is being accumulated under the old tiling: results never depend on a knob, so the sums move with the tiling --
        // gathered under the old bands, scattered under the new ones (the checkpoint path); data.pixels is re-packed from them.
        std::vector<float> frame((size_t)g->width * g->height * 4);
        int rc = GroupReadAccumulator(ctx, frame.data(), frame.size());
        if (rc != CGPT_OK) return rc;
        const uint32_t old_cfg = g->band_rows_cfg;
        g->band_rows_cfg = value;
        cgpt_render_params p{};
        p.width = g->width; p.height = g->height; p.row_begin = 0; p.row_end = g->height;
        rc = GroupWriteAccumulator(ctx, &p, frame.data(), frame.size(), g->num_accumulated);
        if (rc != CGPT_OK) g->band_rows_cfg = old_cfg;
        return rc;
    }
    for (cgpt_ctx* m : g->members) { const int rc = cgpt_set_tuning(m, name, value); if (rc != CGPT_OK) return Propagate(ctx, m, rc); }
    return CGPT_OK;
}

int GroupSynchronize(cgpt_ctx* ctx)
{
    for (cgpt_ctx* m : ctx->group->members) { const int rc = cgpt_synchronize(m); if (rc != CGPT_OK) return Propagate(ctx, m, rc); }
    return CGPT_OK;
}

}  // namespace cgpt
