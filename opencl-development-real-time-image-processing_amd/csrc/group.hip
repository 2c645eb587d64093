// group.hip — mi355_group_*: one batch of frames sharded over several GPUs (include/mi355_imgfilter.h, "device
// group").  Host code only: one mi355_ctx and one worker thread per member, contiguous frame ranges, no collective;
// every member drives its own GPU through the single-device C-ABI (capi.hip).  The reference has one queue on one
// device (RT/src/ProgramHandler.cpp:108) — this is the product-level form of the batched-frame mode BASELINE.json's
// north_star names (bench.py shards the same way over processes).
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <new>
#include <thread>
#include <vector>

#include "../../include/mi355_imgfilter.h"

int mi355_internal_install_generated(mi355_ctx* ctx, int k, float sigma, const float* w_k2);  // capi.hip

namespace {

struct Member {
    int device = 0;
    mi355_ctx* ctx = nullptr;
    int status = MI355_OK;
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    std::function<int(Member&)> job;
    bool has_job = false, done = false, quit = false;

    void loop()
    {
        std::unique_lock<std::mutex> lk(m);
        for (;;) {
            cv.wait(lk, [&] { return has_job || quit; });
            if (quit && !has_job)
                return;
            auto fn = std::move(job);
            has_job = false;
            lk.unlock();
            const int rc = fn(*this);
            lk.lock();
            status = rc;
            done = true;
            cv.notify_all();
        }
    }
    void post(std::function<int(Member&)> fn)
    {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(fn);
        has_job = true;
        done = false;
        cv.notify_all();
    }
    int wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [&] { return done; });
        return status;
    }
};

bool filter_needs_gauss(int f) { return f == MI355_FILTER_GAUSS || f == MI355_FILTER_PIPELINE; }

uint32_t fbits(float f)
{
    uint32_t u;
    std::memcpy(&u, &f, 4);
    return u;
}

}  // namespace

struct mi355_group {
    std::vector<std::unique_ptr<Member>> members;
    std::mutex call;  // one group call at a time
    int input_format = MI355_INPUT_RGBA;
    std::vector<std::pair<int, uint32_t>> keys;  // (k, sigma bits) whose table every member already holds

    // run fn on every member's worker at once; first failing member's code in member order
    int all(const std::function<int(Member&, int)>& fn)
    {
        for (size_t i = 0; i < members.size(); i++)
            members[i]->post([&fn, i](Member& mb) { return fn(mb, (int)i); });
        int rc = MI355_OK;
        for (auto& mb : members) {
            const int r = mb->wait();
            if (rc == MI355_OK && r != MI355_OK)
                rc = r;
        }
        return rc;
    }

    // the (k, sigma) table: generated once, the same bytes installed on every member
    int ensure_table(int k, float sigma)
    {
        for (auto& e : keys)
            if (e.first == k && e.second == fbits(sigma))
                return MI355_OK;
        std::vector<float> tab((size_t)(k > 0 ? k * k : 1));
        int rc = mi355_gauss_weights(k, sigma, tab.data());
        if (rc != MI355_OK)
            return rc;
        rc = all([&](Member& mb, int) { return mi355_internal_install_generated(mb.ctx, k, sigma, tab.data()); });
        if (rc == MI355_OK) {
            if (keys.size() >= 16)  // the members keep the 16 most recent generated tables; so does this list
                keys.erase(keys.begin());
            keys.emplace_back(k, fbits(sigma));
        }
        return rc;
    }
};

extern "C" {

#define MI355_API __attribute__((visibility("default")))

MI355_API int mi355_group_shard(int member, int nmembers, int nframes, int* first_frame, int* count)
{
    if (nmembers <= 0 || member < 0 || member >= nmembers || nframes < 0 || !first_frame || !count)
        return MI355_ERR_BAD_ARG;
    const int base = nframes / nmembers, rem = nframes % nmembers;
    *first_frame = member * base + (member < rem ? member : rem);
    *count = base + (member < rem ? 1 : 0);
    return MI355_OK;
}

MI355_API int mi355_group_destroy(mi355_group* g)
{
    if (!g)
        return MI355_ERR_BAD_ARG;
    for (auto& mb : g->members) {
        if (mb->th.joinable()) {
            mb->post([](Member& m) {
                const int rc = m.ctx ? mi355_ctx_destroy(m.ctx) : MI355_OK;
                m.ctx = nullptr;
                return rc;
            });
            mb->wait();
            {
                std::lock_guard<std::mutex> lk(mb->m);
                mb->quit = true;
                mb->cv.notify_all();
            }
            mb->th.join();
        }
    }
    delete g;
    return MI355_OK;
}

MI355_API int mi355_group_create(int ndev, const int* devices, mi355_group** out)
{
    if (!out)
        return MI355_ERR_BAD_ARG;
    *out = nullptr;
    if (ndev <= 0 || ndev > 64)
        return MI355_ERR_BAD_ARG;
    int avail = 0;
    if (mi355_device_count(&avail) != MI355_OK || avail <= 0)
        return MI355_ERR_NO_DEVICE;
    for (int i = 0; i < ndev; i++) {
        const int d = devices ? devices[i] : i;
        if (d < 0 || d >= avail)
            return MI355_ERR_NO_DEVICE;
    }
    mi355_group* g = new (std::nothrow) mi355_group();
    if (!g)
        return MI355_ERR_NOMEM;
    try {  // nothing may throw across the C boundary: a failed allocation or thread start is an error code
        for (int i = 0; i < ndev; i++) {
            auto mb = std::make_unique<Member>();
            mb->device = devices ? devices[i] : i;
            Member* raw = mb.get();
            mb->th = std::thread([raw] { raw->loop(); });
            g->members.push_back(std::move(mb));
        }
    } catch (...) {
        mi355_group_destroy(g);
        return MI355_ERR_NOMEM;
    }
    // every member creates its context on its own worker thread (the thread that will drive that GPU)
    const int rc = g->all([](Member& mb, int) { return mi355_ctx_create(mb.device, &mb.ctx); });
    if (rc != MI355_OK) {
        mi355_group_destroy(g);
        return rc;
    }
    *out = g;
    return MI355_OK;
}

MI355_API int mi355_group_size(mi355_group* g, int* ndev)
{
    if (!g || !ndev)
        return MI355_ERR_BAD_ARG;
    *ndev = (int)g->members.size();
    return MI355_OK;
}

MI355_API int mi355_group_member_ctx(mi355_group* g, int member, mi355_ctx** ctx)
{
    if (!g || !ctx || member < 0 || member >= (int)g->members.size())
        return MI355_ERR_BAD_ARG;
    *ctx = g->members[(size_t)member]->ctx;
    return MI355_OK;
}

MI355_API int mi355_group_member_status(mi355_group* g, int member)
{
    if (!g || member < 0 || member >= (int)g->members.size())
        return MI355_ERR_BAD_ARG;
    return g->members[(size_t)member]->status;
}

MI355_API int mi355_group_set_gauss_mode(mi355_group* g, int mode)
{
    if (!g)
        return MI355_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->call);
    return g->all([&](Member& mb, int) { return mi355_ctx_set_gauss_mode(mb.ctx, mode); });
}

MI355_API int mi355_group_set_impl(mi355_group* g, int impl)
{
    if (!g)
        return MI355_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->call);
    return g->all([&](Member& mb, int) { return mi355_ctx_set_impl(mb.ctx, impl); });
}

MI355_API int mi355_group_set_input_format(mi355_group* g, int format)
{
    if (!g)
        return MI355_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->call);
    const int rc = g->all([&](Member& mb, int) { return mi355_ctx_set_input_format(mb.ctx, format); });
    if (rc == MI355_OK)
        g->input_format = format;
    return rc;
}

MI355_API int mi355_group_set_gauss_weights(mi355_group* g, int k, float sigma, const float* w_k2)
{
    if (!g || !w_k2)
        return MI355_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->call);
    const int rc = g->all([&](Member& mb, int) { return mi355_ctx_set_gauss_weights(mb.ctx, k, sigma, w_k2); });
    if (rc == MI355_OK) {
        bool known = false;
        for (auto& e : g->keys)
            known = known || (e.first == k && e.second == fbits(sigma));
        if (!known)
            g->keys.emplace_back(k, fbits(sigma));
    }
    return rc;
}

MI355_API int mi355_group_filter_batched(mi355_group* g, int filter, const uint8_t* rgba, uint8_t* out, int w, int h,
                                         int nframes, int k, float sigma, double* elapsed_ms)
{
    if (!g || !rgba || !out || w <= 0 || h <= 0 || nframes <= 0)
        return MI355_ERR_BAD_ARG;
    const int bpp = mi355_filter_out_bpp(filter);
    if (bpp < 0)
        return MI355_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->call);
    if (filter_needs_gauss(filter)) {
        const int rc = g->ensure_table(k, sigma);
        if (rc != MI355_OK)
            return rc;
    }
    const size_t fpx = (size_t)w * h, in_bpp = g->input_format == MI355_INPUT_BGR ? 3 : 4;
    const int n = (int)g->members.size();
    const auto t0 = std::chrono::steady_clock::now();
    const int rc = g->all([&](Member& mb, int i) {
        int first = 0, count = 0;
        mi355_group_shard(i, n, nframes, &first, &count);
        if (count == 0)
            return (int)MI355_OK;
        return mi355_filter_stream(mb.ctx, filter, rgba + (size_t)first * fpx * in_bpp, out + (size_t)first * fpx * (size_t)bpp,
                                   w, h, count, 0, k, sigma, nullptr);
    });
    if (elapsed_ms)
        *elapsed_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
    return rc;
}

MI355_API int mi355_group_filter_dev(mi355_group* g, int filter, const void* const* d_in, void* const* d_out, int w, int h,
                                     const int* nframes, int k, float sigma)
{
    if (!g || !d_in || !d_out || !nframes || mi355_filter_out_bpp(filter) < 0)
        return MI355_ERR_BAD_ARG;
    for (size_t i = 0; i < g->members.size(); i++)
        if (nframes[i] < 0)
            return MI355_ERR_BAD_ARG;
    std::lock_guard<std::mutex> lk(g->call);
    if (filter_needs_gauss(filter)) {
        const int rc = g->ensure_table(k, sigma);
        if (rc != MI355_OK)
            return rc;
    }
    return g->all([&](Member& mb, int i) {
        if (nframes[i] == 0)
            return (int)MI355_OK;
        const int rc = mi355_filter_dev(mb.ctx, filter, d_in[i], d_out[i], w, h, nframes[i], k, sigma);
        return rc != MI355_OK ? rc : mi355_sync(mb.ctx);
    });
}

}  // extern "C"
