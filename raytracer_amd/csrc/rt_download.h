// Device-to-caller copies of rendered frames, shared by the single-GPU entry points (rt_render,
// rt_render_u8, rt_render_ppm) and the several-GPU ones (rt_render_multi*).
//
// The reference's render returns a Canvas in host memory (src/lib.rs:118-126); here the frame is
// produced in HBM and has to cross PCIe.  A 3840x2160 f64 frame is 199 MB, several times the render
// time, so the copy is the part of the boundary worth engineering:
//   * a destination that is already page-locked (rt_host_alloc, hipHostMalloc, hipHostRegister) is
//     written by the copy engine directly;
//   * any other destination is pageable memory the GPU cannot reach: the frame then goes through a
//     small ring of page-locked slots, and a pool of host threads moves slot k to its place while
//     the copy engine fills slot k + 1 (a plain hipMemcpy to pageable memory does the same thing
//     with one slot and one thread).
// A rank's buffer is compact: its k-th 8-row tile sits at k * tile_bytes and belongs at
// (k * world_size + rank) * tile_bytes of the frame; with world_size 1 that is one contiguous copy.
#ifndef RT_DOWNLOAD_H
#define RT_DOWNLOAD_H

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <thread>
#include <vector>

namespace rtdl {

// A few persistent host threads that split memcpy-like jobs.  Callers on different threads may use
// it at the same time (the several-GPU path downloads every rank's tiles concurrently).
class Pool {
public:
    static Pool& get() {
        static Pool p;
        return p;
    }
    int threads() const { return (int)workers_.size() + 1; }
    // Runs fn(0) .. fn(parts - 1), the caller's thread taking part, and returns when all are done.
    void run(int parts, const std::function<void(int)>& fn) {
        if (parts <= 0) return;
        if (parts == 1 || workers_.empty()) {
            for (int i = 0; i < parts; ++i) fn(i);
            return;
        }
        Job job{&fn, parts, 0, 0};
        {
            std::lock_guard<std::mutex> lock(mu_);
            jobs_.push_back(&job);  // a queued job always has parts left to start
        }
        cv_.notify_all();
        for (;;) {
            int part;
            {
                std::lock_guard<std::mutex> lock(mu_);
                if (job.next >= job.parts) break;
                part = claim(&job);
            }
            fn(part);
            std::lock_guard<std::mutex> lock(mu_);
            ++job.finished;
        }
        std::unique_lock<std::mutex> lock(mu_);
        done_cv_.wait(lock, [&] { return job.finished == job.parts; });
    }
    ~Pool() {
        {
            std::lock_guard<std::mutex> lock(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }

private:
    struct Job {
        const std::function<void(int)>* fn;
        int parts, next, finished;
    };
    Pool() {
        int n = 8;
        if (const char* e = getenv("RT_COPY_THREADS")) n = atoi(e);
        const int hw = (int)std::thread::hardware_concurrency();
        if (hw > 0 && n > hw) n = hw;
        if (n < 1) n = 1;
        for (int i = 1; i < n; ++i) workers_.emplace_back([this] { loop(); });
    }
    // With mu_ held: takes the next part of a queued job and unqueues the job with its last part.
    // A job whose part is claimed cannot finish (and its owner cannot return) before that part is
    // reported done, so the pointer stays valid until then.
    int claim(Job* job) {
        const int part = job->next++;
        if (job->next >= job->parts) jobs_.erase(std::remove(jobs_.begin(), jobs_.end(), job), jobs_.end());
        return part;
    }
    void loop() {
        for (;;) {
            Job* job;
            int part;
            {
                std::unique_lock<std::mutex> lock(mu_);
                cv_.wait(lock, [&] { return stop_ || !jobs_.empty(); });
                if (stop_) return;
                job = jobs_.front();
                part = claim(job);
            }
            (*job->fn)(part);
            bool last;
            {
                std::lock_guard<std::mutex> lock(mu_);
                last = ++job->finished == job->parts;  // the job may be gone after this
            }
            if (last) done_cv_.notify_all();
        }
    }
    std::mutex mu_;
    std::condition_variable cv_, done_cv_;
    std::deque<Job*> jobs_;
    std::vector<std::thread> workers_;
    bool stop_ = false;
};

// True when the GPU can write `p` directly (page-locked host memory known to the HIP runtime).
inline bool is_pinned(const void* p) {
    hipPointerAttribute_t attr;
    std::memset(&attr, 0, sizeof attr);
    if (hipPointerGetAttributes(&attr, p) != hipSuccess) {
        (void)hipGetLastError();  // ordinary memory: the query fails by design, not an error of ours
        return false;
    }
    return attr.type == hipMemoryTypeHost;
}

class Downloader {
public:
    static constexpr int SLOTS = 4;
    static constexpr size_t SLOT_BYTES = (size_t)8 << 20;

    // Creates the copy stream on `device`; the slots are allocated when a pageable destination
    // first needs them.
    hipError_t init(int device) {
        device_ = device;
        hipError_t e = hipSetDevice(device);
        if (e != hipSuccess) return e;
        if ((e = hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking)) != hipSuccess) return e;
        for (int i = 0; i < SLOTS; ++i)
            if ((e = hipEventCreateWithFlags(&ev_[i], hipEventDisableTiming)) != hipSuccess) return e;
        return hipSuccess;
    }
    void destroy() {
        if (device_ < 0) return;
        (void)hipSetDevice(device_);
        for (int i = 0; i < SLOTS; ++i) {
            if (slot_[i]) (void)hipHostFree(slot_[i]);
            if (ev_[i]) (void)hipEventDestroy(ev_[i]);
            slot_[i] = nullptr;
            ev_[i] = nullptr;
        }
        if (stream_) (void)hipStreamDestroy(stream_);
        stream_ = nullptr;
        device_ = -1;
    }
    hipStream_t stream() const { return stream_; }

    // Queues a copy of `n` contiguous bytes to page-locked memory behind `ready` and returns at once;
    // finish() waits for everything queued.
    hipError_t enqueue(const char* d_src, size_t n, char* pinned_dst, hipEvent_t ready) {
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return e;
        if (ready && (e = hipStreamWaitEvent(stream_, ready, 0)) != hipSuccess) return e;
        return hipMemcpyAsync(pinned_dst, d_src, n, hipMemcpyDeviceToHost, stream_);
    }
    hipError_t finish() { return hipStreamSynchronize(stream_); }

    // Copies `total` bytes of a rank's compact buffer on the device to their places in the frame at
    // `dst` (host).  `ready`: an event after which the buffer holds the frame (may be null).
    // Returns when every byte has landed.  *staged tells which way the bytes went.
    hipError_t download(const char* d_src, size_t total, size_t tile_bytes, int rank, int world, char* dst,
                        hipEvent_t ready, bool* staged = nullptr) {
        hipError_t e = hipSetDevice(device_);
        if (e != hipSuccess) return e;
        if (staged) *staged = false;
        if (total == 0) return hipSuccess;
        if (ready && (e = hipStreamWaitEvent(stream_, ready, 0)) != hipSuccess) return e;
        if (tile_bytes == 0 || world <= 1) {
            tile_bytes = total;
            world = 1;
            rank = 0;
        }
        if (is_pinned(dst)) {
            const size_t full = total / tile_bytes, tail = total - full * tile_bytes;
            if (world == 1) {
                e = hipMemcpyAsync(dst, d_src, total, hipMemcpyDeviceToHost, stream_);
            } else {
                e = hipSuccess;
                if (full)
                    e = hipMemcpy2DAsync(dst + (size_t)rank * tile_bytes, (size_t)world * tile_bytes, d_src, tile_bytes,
                                         tile_bytes, full, hipMemcpyDeviceToHost, stream_);
                if (e == hipSuccess && tail)
                    e = hipMemcpyAsync(dst + (full * world + rank) * tile_bytes, d_src + full * tile_bytes, tail,
                                       hipMemcpyDeviceToHost, stream_);
            }
            if (e != hipSuccess) return e;
            return hipStreamSynchronize(stream_);
        }
        if (staged) *staged = true;
        const size_t chunk = SLOT_BYTES;
        for (int i = 0; i < SLOTS; ++i)  // (each slot on its own: an earlier call may have got only some of them)
            if (!slot_[i] && (e = hipHostMalloc((void**)&slot_[i], chunk, hipHostMallocDefault)) != hipSuccess) {
                slot_[i] = nullptr;
                return e;
            }
        const size_t n_chunks = (total + chunk - 1) / chunk;
        // On an error nothing may stay in flight into the slots: the next call reuses them.
        auto drained = [&](hipError_t err) {
            (void)hipStreamSynchronize(stream_);
            return err;
        };
        auto place = [&](size_t c) -> hipError_t {
            const int s = (int)(c % SLOTS);
            hipError_t pe = hipEventSynchronize(ev_[s]);
            if (pe != hipSuccess) return pe;
            const size_t off = c * chunk, n = std::min(chunk, total - off);
            scatter(slot_[s], off, n, tile_bytes, rank, world, dst);
            return hipSuccess;
        };
        for (size_t c = 0; c < n_chunks; ++c) {
            if (c >= (size_t)SLOTS && (e = place(c - SLOTS)) != hipSuccess) return drained(e);  // frees the slot
            const int s = (int)(c % SLOTS);
            const size_t off = c * chunk, n = std::min(chunk, total - off);
            if ((e = hipMemcpyAsync(slot_[s], d_src + off, n, hipMemcpyDeviceToHost, stream_)) != hipSuccess) return drained(e);
            if ((e = hipEventRecord(ev_[s], stream_)) != hipSuccess) return drained(e);
        }
        for (size_t c = n_chunks > (size_t)SLOTS ? n_chunks - SLOTS : 0; c < n_chunks; ++c)
            if ((e = place(c)) != hipSuccess) return drained(e);
        return hipSuccess;
    }

private:
    // Bytes [off, off + n) of the compact buffer, now at `src`, to their places in the frame.
    static void scatter(const char* src, size_t off, size_t n, size_t tile_bytes, int rank, int world, char* dst) {
        Pool& pool = Pool::get();
        const size_t grain = (size_t)512 << 10;
        int parts = (int)std::min<size_t>((size_t)pool.threads(), (n + grain - 1) / grain);
        if (parts < 1) parts = 1;
        const size_t per = ((n + parts - 1) / parts + 63) & ~(size_t)63;
        pool.run(parts, [&](int p) {
            size_t a = (size_t)p * per, b = std::min(n, a + per);
            while (a < b) {
                const size_t o = off + a, tile = o / tile_bytes, within = o - tile * tile_bytes;
                const size_t len = std::min(b - a, tile_bytes - within);
                std::memcpy(dst + (tile * world + rank) * tile_bytes + within, src + a, len);
                a += len;
            }
        });
    }
    int device_ = -1;
    hipStream_t stream_ = nullptr;
    char* slot_[SLOTS] = {};
    hipEvent_t ev_[SLOTS] = {};
};

inline double now_ms() {
    return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace rtdl
#endif  // RT_DOWNLOAD_H
