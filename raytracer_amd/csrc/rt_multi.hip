// One frame on several GPUs from ONE host call: the replacement of the reference's render
// (src/lib.rs:77-133: twelve worker threads over a cyclic pixel partition, one channel back to the
// main thread) when the host process owns several MI355X.
//
// An rt_multi holds one copy of the scene per rank (rt_scene_create on that rank's GPU), a render
// stream and a compact tile buffer per rank.  A frame is rendered in cyclic 8-row tiles, tile t on
// rank t % n (the partition of rt_abi.h); no rank talks to another while rendering.  The reference's
// channel (lib.rs:102-126) becomes one of two exchanges:
//   * rt_render_multi / rt_render_multi_u8 (frame to HOST memory): every GPU copies its own tiles
//     over its own PCIe link straight to their rows of the caller's frame.  Nothing funnels through
//     one GPU: n links carry the frame instead of one, and xGMI is not used at all.
//   * rt_render_multi_device (frame stays in HBM on rank 0's GPU): one kernel on rank 0's GPU reads
//     every peer's compact buffer through peer-mapped pointers and writes the rows in frame order.
//     Each peer's bytes come over that peer's own xGMI link, all links at once.  (Without peer access
//     the buffers are first copied to rank 0 with hipMemcpyPeerAsync.)
// Built only on the public single-GPU entry points of rt_abi.h plus the HIP runtime.
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "rt_abi.h"
#include "rt_download.h"

extern "C" void rt_internal_set_error(const char* message);  // rt_amd.hip

namespace {

int mfail(int code, const std::string& msg) {
    rt_internal_set_error(msg.c_str());  // one error channel for the whole library: rt_last_error()
    return code;
}
#define MHIP(expr)                                                                                  \
    do {                                                                                            \
        hipError_t e_ = (expr);                                                                     \
        if (e_ != hipSuccess) return mfail(RT_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

constexpr int MAX_RANKS = 64;
struct GatherArgs {
    const uint32_t* src[MAX_RANKS];  // every rank's compact buffer, as seen from rank 0's GPU
    unsigned long long bytes[MAX_RANKS];  // its length
};

// Frame assembly on rank 0's GPU: word w of rank r's compact buffer lies in that rank's tile
// k = w / tile_words and goes to frame tile k * n + r.  blockIdx.y is the rank, so one launch pulls
// from every peer at once.  Consecutive lanes read and write consecutive words.  A full tile is a
// whole number of words (8 rows), but a byte frame's short last tile need not be: the one to three
// bytes beyond the last whole word of the rank that owns it are copied one by one.
__global__ void __launch_bounds__(256) gather_tiles_kernel(GatherArgs a, int n_ranks, unsigned long long tile_words,
                                                           uint32_t* __restrict__ frame) {
    const int r = blockIdx.y;
    const unsigned long long total = a.bytes[r] / 4;
    const uint32_t* __restrict__ src = a.src[r];
    for (unsigned long long w = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; w < total;
         w += (unsigned long long)gridDim.x * blockDim.x) {
        const unsigned long long k = w / tile_words, within = w - k * tile_words;
        frame[(k * (unsigned long long)n_ranks + (unsigned long long)r) * tile_words + within] = src[w];
    }
    const unsigned long long tail = total * 4 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 4 && tail < a.bytes[r]) {
        const unsigned long long tile_bytes = tile_words * 4, k = tail / tile_bytes, within = tail - k * tile_bytes;
        reinterpret_cast<uint8_t*>(frame)[(k * (unsigned long long)n_ranks + (unsigned long long)r) * tile_bytes + within] =
            reinterpret_cast<const uint8_t*>(src)[tail];
    }
}

}  // namespace

struct rt_multi {
    int n = 0;
    std::vector<int> device;
    std::vector<rt_scene*> scene;
    std::vector<hipStream_t> stream;
    std::vector<hipEvent_t> done;
    std::vector<void*> d_buf;
    std::vector<size_t> buf_bytes;
    std::vector<rtdl::Downloader> dl;
    // device gather (rank 0's GPU)
    std::vector<char> peer_ok;     // rank 0's GPU can read rank r's memory directly
    std::vector<void*> d_stage;    // copies of the peers' buffers on rank 0's GPU when it cannot
    std::vector<size_t> stage_bytes;
    hipEvent_t g0 = nullptr, g1 = nullptr;  // around the gather, for its device time
};

namespace {

struct FrameShape {
    size_t tile_bytes;                 // one full 8-row tile of the frame
    std::vector<size_t> rank_bytes;    // each rank's compact buffer
    size_t frame_bytes;
};

int frame_shape(const rt_multi* m, const rt_camera* cam, size_t pixel_bytes, FrameShape& f) {
    if (cam->width <= 0 || cam->height <= 0) return mfail(RT_ERR_INVALID, "empty canvas");
    const size_t row = (size_t)cam->width * pixel_bytes;
    f.tile_bytes = row * RT_TILE;
    f.frame_bytes = row * (size_t)cam->height;
    f.rank_bytes.assign(m->n, 0);
    for (int r = 0; r < m->n; ++r) {
        rt_params p{0, r, m->n, 0};
        const int64_t rows = rt_local_rows(cam->height, &p);
        if (rows < 0) return mfail(RT_ERR_INVALID, "bad canvas height");
        f.rank_bytes[r] = (size_t)rows * row;
    }
    return RT_OK;
}

// The render stream, "done" event and copy stream of every rank, made on first need.
int rank_streams(rt_multi* m) {
    for (int r = 0; r < m->n; ++r) {
        if (m->stream[r]) continue;
        if (hipSetDevice(m->device[r]) != hipSuccess ||
            hipStreamCreateWithFlags(&m->stream[r], hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&m->done[r], hipEventDisableTiming) != hipSuccess ||
            m->dl[r].init(m->device[r]) != hipSuccess)
            return mfail(RT_ERR_HIP, "stream or event creation failed");
    }
    return RT_OK;
}

int reserve(rt_multi* m, int r, size_t bytes) {
    if (bytes <= m->buf_bytes[r]) return RT_OK;
    MHIP(hipSetDevice(m->device[r]));
    if (m->d_buf[r]) (void)hipFree(m->d_buf[r]);
    m->d_buf[r] = nullptr;
    m->buf_bytes[r] = 0;
    MHIP(hipMalloc(&m->d_buf[r], bytes));
    m->buf_bytes[r] = bytes;
    return RT_OK;
}

// Launches every rank's share of the frame on its GPU and records `done[r]` behind it.
int launch_all(rt_multi* m, const rt_camera* cam, int max_depth, bool u8, const FrameShape& f) {
    for (int r = 0; r < m->n; ++r) {
        int rc = reserve(m, r, f.rank_bytes[r]);
        if (rc != RT_OK) return rc;
    }
    for (int r = 0; r < m->n; ++r) {
        rt_params p{max_depth, r, m->n, 0};
        int rc = u8 ? rt_render_device_u8(m->scene[r], cam, &p, (uint8_t*)m->d_buf[r], m->stream[r])
                    : rt_render_device(m->scene[r], cam, &p, (double*)m->d_buf[r], m->stream[r]);
        if (rc != RT_OK) return rc;
        MHIP(hipSetDevice(m->device[r]));
        MHIP(hipEventRecord(m->done[r], m->stream[r]));
    }
    return RT_OK;
}

int collect_stats(rt_multi* m, rt_multi_stats* st, double t0, double t_render, double t_end) {
    if (!st) return RT_OK;
    std::memset(st, 0, sizeof *st);
    for (int r = 0; r < m->n; ++r) {
        rt_stats s{};
        if (rt_scene_ray_counts(m->scene[r], &s) != RT_OK) return RT_ERR_HIP;
        st->rays_primary += s.rays_primary;
        st->rays_shadow += s.rays_shadow;
        st->rays_reflect += s.rays_reflect;
        st->rays_refract += s.rays_refract;
        if (s.kernel_ms > st->kernel_ms) st->kernel_ms = s.kernel_ms;
    }
    st->render_wall_ms = (float)(t_render - t0);
    st->exchange_ms = (float)(t_end - t_render);
    st->total_ms = (float)(t_end - t0);
    return RT_OK;
}

int render_to_host(rt_multi* m, const rt_camera* cam, int max_depth, void* out, bool u8, rt_multi_stats* st) {
    if (!m || !cam) return mfail(RT_ERR_INVALID, "null handle or camera");
    if (max_depth < 0 || max_depth > RT_MAX_DEPTH) return mfail(RT_ERR_LIMIT, "max_depth outside 0..RT_MAX_DEPTH");
    FrameShape f;
    int rc = frame_shape(m, cam, u8 ? 3 : 3 * sizeof(double), f);
    if (rc != RT_OK) return rc;
    if (!out) return mfail(RT_ERR_INVALID, "null output buffer");
    const double t0 = rtdl::now_ms();
    if (m->n == 1) {
        // One rank is the single-GPU entry point, which renders large frames in bands and sends every
        // band home while the next one renders.
        rt_params p{max_depth, 0, 1, 0};
        rt_stats one{};
        rc = u8 ? rt_render_u8(m->scene[0], cam, &p, static_cast<uint8_t*>(out), &one)
                : rt_render(m->scene[0], cam, &p, static_cast<double*>(out), &one);
        if (rc != RT_OK) return rc;
        const double t_end = rtdl::now_ms();
        if (st) {
            std::memset(st, 0, sizeof *st);
            st->rays_primary = one.rays_primary;
            st->rays_shadow = one.rays_shadow;
            st->rays_reflect = one.rays_reflect;
            st->rays_refract = one.rays_refract;
            st->kernel_ms = one.kernel_ms;
            st->total_ms = (float)(t_end - t0);
            st->render_wall_ms = one.kernel_ms < st->total_ms ? one.kernel_ms : st->total_ms;
            st->exchange_ms = st->total_ms - st->render_wall_ms;
        }
        return RT_OK;
    }
    if ((rc = launch_all(m, cam, max_depth, u8, f)) != RT_OK) return rc;
    // Every rank's tiles go home over that GPU's own PCIe link, all ranks at once.  A pinned
    // destination only needs the copies enqueued; a pageable one is filled by host threads through
    // each rank's ring, one driving thread per rank.
    std::vector<hipError_t> err(m->n, hipSuccess);
    auto fetch = [&](int r) {
        err[r] = m->dl[r].download(static_cast<const char*>(m->d_buf[r]), f.rank_bytes[r], f.tile_bytes, r, m->n,
                                   static_cast<char*>(out), m->done[r]);
    };
    double t_render = 0.0;
    if (m->n == 1) {
        fetch(0);
    } else {
        std::vector<std::thread> th;
        for (int r = 1; r < m->n; ++r) th.emplace_back(fetch, r);
        fetch(0);
        for (auto& t : th) t.join();
    }
    for (int r = 0; r < m->n; ++r)
        if (err[r] != hipSuccess) return mfail(RT_ERR_HIP, std::string("frame download: ") + hipGetErrorString(err[r]));
    const double t_end = rtdl::now_ms();
    // the renders overlap the downloads of the ranks that finish first; their own time is the events'
    float longest = 0.0f;
    for (int r = 0; r < m->n; ++r) {
        float ms = 0.0f;
        if (rt_scene_launch_times(m->scene[r], &ms, 1) == 1 && ms > longest) longest = ms;
    }
    t_render = t0 + longest;
    if (t_render > t_end) t_render = t_end;
    return collect_stats(m, st, t0, t_render, t_end);
}

}  // namespace

extern "C" {

int rt_multi_create(const rt_scene_desc* desc, int32_t n_ranks, const int32_t* devices, rt_multi** out) {
    if (!out) return mfail(RT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (n_ranks < 1 || n_ranks > MAX_RANKS) return mfail(RT_ERR_INVALID, "n_ranks outside 1..64");
    const int ndev = rt_device_count();
    if (ndev == 0) {
        // still validate the tables, so that a malformed scene is reported as such without a GPU
        rt_scene* probe = nullptr;
        int rc = rt_scene_create(desc, 0, &probe);
        if (probe) rt_scene_destroy(probe);
        return rc == RT_OK ? mfail(RT_ERR_NO_DEVICE, "no HIP device is visible") : rc;
    }
    rt_multi* m = new rt_multi();
    m->n = n_ranks;
    m->device.resize(n_ranks);
    for (int r = 0; r < n_ranks; ++r) {
        m->device[r] = devices ? devices[r] : r;
        if (m->device[r] < 0 || m->device[r] >= ndev) {
            delete m;
            return mfail(RT_ERR_INVALID, "device ordinal out of range (more ranks than GPUs and no device list?)");
        }
    }
    m->scene.assign(n_ranks, nullptr);
    m->stream.assign(n_ranks, nullptr);
    m->done.assign(n_ranks, nullptr);
    m->d_buf.assign(n_ranks, nullptr);
    m->buf_bytes.assign(n_ranks, 0);
    m->dl.resize(n_ranks);
    m->peer_ok.assign(n_ranks, 0);
    m->d_stage.assign(n_ranks, nullptr);
    m->stage_bytes.assign(n_ranks, 0);
    auto bail = [&](int code, const std::string& msg) {
        rt_multi_destroy(m);
        return mfail(code, msg);
    };
    for (int r = 0; r < n_ranks; ++r) {
        int rc = rt_scene_create(desc, m->device[r], &m->scene[r]);
        if (rc != RT_OK) return bail(rc, std::string(rt_last_error()));
    }
    // One rank renders through the single-GPU entry points, which have streams of their own: the per-rank
    // streams are then only made if the device gather is used.  (Streams share a process's four hardware
    // queues; every idle one more makes it likelier that a frame's copies queue up in front of its renders.)
    if (n_ranks > 1) {
        int rc = rank_streams(m);
        if (rc != RT_OK) return bail(rc, std::string(rt_last_error()));
    }
    // rank 0's GPU reads its peers' buffers directly where the hardware allows it (xGMI)
    if (hipSetDevice(m->device[0]) != hipSuccess || hipEventCreate(&m->g0) != hipSuccess || hipEventCreate(&m->g1) != hipSuccess)
        return bail(RT_ERR_HIP, "event creation failed");
    for (int r = 0; r < n_ranks; ++r) {
        if (m->device[r] == m->device[0]) {
            m->peer_ok[r] = 1;
            continue;
        }
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, m->device[0], m->device[r]) == hipSuccess && can) {
            hipError_t e = hipDeviceEnablePeerAccess(m->device[r], 0);
            if (e == hipSuccess || e == hipErrorPeerAccessAlreadyEnabled) m->peer_ok[r] = 1;
            (void)hipGetLastError();
        }
    }
    *out = m;
    return RT_OK;
}

void rt_multi_destroy(rt_multi* m) {
    if (!m) return;
    for (int r = 0; r < m->n; ++r) {
        if (r < (int)m->device.size()) (void)hipSetDevice(m->device[r]);
        if (m->stream[r]) (void)hipStreamSynchronize(m->stream[r]);
        m->dl[r].destroy();
        if (m->scene[r]) rt_scene_destroy(m->scene[r]);
        if (m->d_buf[r]) (void)hipFree(m->d_buf[r]);
        if (m->done[r]) (void)hipEventDestroy(m->done[r]);
        if (m->stream[r]) (void)hipStreamDestroy(m->stream[r]);
    }
    if (!m->device.empty()) (void)hipSetDevice(m->device[0]);
    for (void* p : m->d_stage)
        if (p) (void)hipFree(p);
    if (m->g0) (void)hipEventDestroy(m->g0);
    if (m->g1) (void)hipEventDestroy(m->g1);
    delete m;
}

int32_t rt_multi_ranks(const rt_multi* m) { return m ? m->n : 0; }

int rt_render_multi(rt_multi* m, const rt_camera* cam, int32_t max_depth, double* rgb_out, rt_multi_stats* stats) {
    return render_to_host(m, cam, max_depth, rgb_out, false, stats);
}

int rt_render_multi_u8(rt_multi* m, const rt_camera* cam, int32_t max_depth, uint8_t* rgb8_out, rt_multi_stats* stats) {
    return render_to_host(m, cam, max_depth, rgb8_out, true, stats);
}

int rt_render_multi_device(rt_multi* m, const rt_camera* cam, int32_t max_depth, void* d_frame, int32_t u8,
                           rt_multi_stats* stats) {
    if (!m || !cam) return mfail(RT_ERR_INVALID, "null handle or camera");
    if (max_depth < 0 || max_depth > RT_MAX_DEPTH) return mfail(RT_ERR_LIMIT, "max_depth outside 0..RT_MAX_DEPTH");
    FrameShape f;
    int rc = frame_shape(m, cam, u8 ? 3 : 3 * sizeof(double), f);
    if (rc != RT_OK) return rc;
    if (!d_frame) return mfail(RT_ERR_INVALID, "null output buffer");
    if ((f.tile_bytes & 3) != 0 || (reinterpret_cast<uintptr_t>(d_frame) & 3) != 0)
        return mfail(RT_ERR_INVALID, "rt_render_multi_device: width * bytes per pixel * 8 and the frame address must be multiples of 4");
    if ((rc = rank_streams(m)) != RT_OK) return rc;
    const double t0 = rtdl::now_ms();
    if ((rc = launch_all(m, cam, max_depth, u8 != 0, f)) != RT_OK) return rc;
    // the exchange runs on rank 0's render stream, behind rank 0's own render
    MHIP(hipSetDevice(m->device[0]));
    hipStream_t s0 = m->stream[0];
    GatherArgs args{};
    for (int r = 0; r < m->n; ++r) {
        if (r > 0) MHIP(hipStreamWaitEvent(s0, m->done[r], 0));
        const void* src = m->d_buf[r];
        if (!m->peer_ok[r] && f.rank_bytes[r]) {
            if (f.rank_bytes[r] > m->stage_bytes[r]) {
                if (m->d_stage[r]) (void)hipFree(m->d_stage[r]);
                m->d_stage[r] = nullptr;
                m->stage_bytes[r] = 0;
                MHIP(hipMalloc(&m->d_stage[r], f.rank_bytes[r]));
                m->stage_bytes[r] = f.rank_bytes[r];
            }
            MHIP(hipMemcpyPeerAsync(m->d_stage[r], m->device[0], m->d_buf[r], m->device[r], f.rank_bytes[r], s0));
            src = m->d_stage[r];
        }
        args.src[r] = static_cast<const uint32_t*>(src);
        args.bytes[r] = f.rank_bytes[r];
    }
    MHIP(hipEventRecord(m->g0, s0));
    size_t most = 0;
    for (size_t b : f.rank_bytes) most = b > most ? b : most;
    if (most) {
        unsigned bx = (unsigned)((most / 4 + 256 * 8 - 1) / (256 * 8));
        if (bx > 65535u * 16u) bx = 65535u * 16u;
        if (bx < 1) bx = 1;
        hipLaunchKernelGGL(gather_tiles_kernel, dim3(bx, (unsigned)m->n), dim3(256), 0, s0, args, m->n,
                           (unsigned long long)(f.tile_bytes / 4), static_cast<uint32_t*>(d_frame));
        MHIP(hipGetLastError());
    }
    MHIP(hipEventRecord(m->g1, s0));
    MHIP(hipStreamSynchronize(s0));
    const double t_end = rtdl::now_ms();
    rc = collect_stats(m, stats, t0, t_end, t_end);
    if (rc != RT_OK) return rc;
    if (stats) {
        float gms = 0.0f;
        MHIP(hipEventElapsedTime(&gms, m->g0, m->g1));
        stats->exchange_ms = gms;
        stats->render_wall_ms = (float)(t_end - t0) - gms;
    }
    return RT_OK;
}

}  // extern "C"
