// Host-side staging of a window's frames (no GPU, no context): the same rectangle of every frame copied into one densely
// packed buffer (page-locked, swk_pinned_alloc) that swk_batch_run then uploads in a single DMA.  The frames of a video are
// cold in the caches (6 MB each at 1080p), so the copy runs at DRAM speed of ONE core when Python does it (numpy slice
// assignment, 0.4-0.6 ms per 21-frame window); here a small persistent pool of threads splits the frames.
#include <stdint.h>
#include <string.h>
#include <atomic>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>
#include "swk.h"

namespace {

struct Job {
    const uint8_t *const *frames = nullptr;
    int count = 0, rows = 0;
    int64_t row_stride = 0, first_byte = 0, row_bytes = 0;
    uint8_t *dst = nullptr;
};

void copy_frames(const Job &j, int f0, int f1)
{
    for (int f = f0; f < f1; ++f) {
        const uint8_t *src = j.frames[f] + j.first_byte;
        uint8_t *d = j.dst + (int64_t)f * j.rows * j.row_bytes;
        for (int r = 0; r < j.rows; ++r) memcpy(d + (int64_t)r * j.row_bytes, src + (int64_t)r * j.row_stride, (size_t)j.row_bytes);
    }
}

class Pool {
public:
    explicit Pool(int workers)
    {
        for (int i = 0; i < workers; ++i) threads_.emplace_back([this, i] { loop(i); });
    }
    ~Pool()
    {
        { std::lock_guard<std::mutex> g(m_); stop_ = true; ++epoch_; }
        cv_.notify_all();
        for (auto &t : threads_) t.join();
    }
    // the caller is worker number `workers`: it copies its share too, then waits for the others
    void run(const Job &j)
    {
        const int parts = (int)threads_.size() + 1;
        { std::lock_guard<std::mutex> g(m_); job_ = j; pending_ = (int)threads_.size(); ++epoch_; }
        cv_.notify_all();
        copy_frames(j, (int)((int64_t)j.count * (parts - 1) / parts), j.count);
        std::unique_lock<std::mutex> g(m_);
        done_.wait(g, [this] { return pending_ == 0; });
    }
private:
    void loop(int idx)
    {
        unsigned seen = 0;
        for (;;) {
            Job j;
            {
                std::unique_lock<std::mutex> g(m_);
                cv_.wait(g, [&] { return epoch_ != seen; });
                seen = epoch_;
                if (stop_) return;
                j = job_;
            }
            const int parts = (int)threads_.size() + 1;
            copy_frames(j, (int)((int64_t)j.count * idx / parts), (int)((int64_t)j.count * (idx + 1) / parts));
            { std::lock_guard<std::mutex> g(m_); --pending_; }
            done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex m_;
    std::condition_variable cv_, done_;
    Job job_;
    unsigned epoch_ = 0;
    int pending_ = 0;
    bool stop_ = false;
};

std::mutex g_pool_mutex;       // one staging call at a time (the pool holds one job)
Pool *g_pool = nullptr;

}  // namespace

#pragma GCC visibility push(default)
extern "C" {

int32_t swk_stage_frames(const uint8_t *const *frames, int32_t count, int64_t row_stride, int32_t y0, int32_t rows, int64_t x_bytes,
                         int64_t row_bytes, uint8_t *dst, int32_t threads)
{
    if (!frames || !dst || count < 1 || rows < 1 || row_bytes < 1 || y0 < 0 || x_bytes < 0 || row_stride < x_bytes + row_bytes)
        return SWK_ERR_ARG;
    for (int f = 0; f < count; ++f)
        if (!frames[f]) return SWK_ERR_ARG;
    Job j;
    j.frames = frames; j.count = count; j.rows = rows; j.row_stride = row_stride;
    j.first_byte = (int64_t)y0 * row_stride + x_bytes; j.row_bytes = row_bytes; j.dst = dst;
    if (threads <= 1 || count < 4 || (int64_t)count * rows * row_bytes < (1 << 20)) { copy_frames(j, 0, count); return SWK_OK; }
    std::lock_guard<std::mutex> g(g_pool_mutex);
    if (!g_pool) g_pool = new Pool(3);          // 3 workers + the caller; never destroyed (process lifetime)
    g_pool->run(j);
    return SWK_OK;
}

// Segment images of a window cut in one go (extract_segment_images, image_filtering.py:338-369, for frames that must not be kept alive
// by their segments: the ROI-stream reader's page-locked blocks are reused): box i = rows [boxes[4i], boxes[4i+1]) x columns
// [boxes[4i+2], boxes[4i+3]) of frame frame_of[i], copied densely to out + offsets[i].
int32_t swk_cut_boxes(const uint8_t *const *frames, int32_t nframes, int64_t row_stride, int32_t pixel_bytes, int32_t count,
                      const int32_t *frame_of, const int32_t *boxes, const int64_t *offsets, uint8_t *out)
{
    if (!frames || !frame_of || !boxes || !offsets || !out || count < 0 || nframes < 1 || pixel_bytes < 1) return SWK_ERR_ARG;
    for (int i = 0; i < count; ++i) {
        const int f = frame_of[i], r0 = boxes[4 * i], r1 = boxes[4 * i + 1], c0 = boxes[4 * i + 2], c1 = boxes[4 * i + 3];
        if (f < 0 || f >= nframes || !frames[f] || r0 < 0 || c0 < 0) return SWK_ERR_ARG;
        if (r1 <= r0 || c1 <= c0) continue;
        const int64_t rowb = (int64_t)(c1 - c0) * pixel_bytes;
        const uint8_t *src = frames[f] + (int64_t)r0 * row_stride + (int64_t)c0 * pixel_bytes;
        uint8_t *dst = out + offsets[i];
        for (int r = 0; r < r1 - r0; ++r) memcpy(dst + (int64_t)r * rowb, src + (int64_t)r * row_stride, (size_t)rowb);
    }
    return SWK_OK;
}

}  // extern "C"
#pragma GCC visibility pop
