// twhost.h — C++ host layer above the C ABI: the job queue, the per-GPU consumers and the result pump of
// tidal-wave (reference: src/message_queue.h, src/consumer.{h,cpp}, src/manager.{h,cpp}), re-stated on
// std::thread / std::mutex instead of libuv primitives.  Everything device-side goes through include/twflow.h.
#pragma once
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/twflow.h"

namespace twhost {

// An image pair that is already decoded, in the caller's memory (8-bit gray, row stride in bytes).  Not part of
// the reference's Request: the throughput driver (tools/bench_queue.cpp, BASELINE config 4) uses it to feed the
// manager queue without touching the file system.  The buffers stay valid until the pair's response arrives;
// page-locked buffers (tw_host_alloc) are DMA-ed from directly.
struct RawPair {
    const uint8_t* expect = nullptr;
    const uint8_t* target = nullptr;
    int width = 0, height = 0;
    ptrdiff_t stride = 0;
};
// /root/reference/src/message_queue.h:13-48
struct Request {
    std::string expect_image;
    std::string target_image;
    double threshold;
    int span;
    RawPair raw;  // raw.expect != nullptr: the paths above are labels only
};
struct Vector {
    int x, y;
    double dx, dy;
};
struct Response {
    std::vector<Vector> vectors;
    std::string expect_image, target_image;
    float time = 0;
    double threshold = 0;
    int span = 0;
    std::string status, reason;
    int width = 0, height = 0;
    long long pushedNs = 0;  // host-internal: when the consumer handed the response to the pump (steady clock)
};
// How long responses sat between a consumer's push and the observer's callback since the last markEpoch() — the one
// pump thread of src/manager.cpp:80-90 is where N consumers' results are serialised
struct PumpStats {
    long delivered = 0;
    double meanUs = 0, maxUs = 0;
    double busyMs = 0;  // time the pump spent inside the observer's callbacks
};
struct Report {
    int requestCount = 0, dataCount = 0, errorCount = 0;
};

// struct Parameter of the reference (src/manager.h): per-instance options resolved by the broker
struct Parameter {
    double threshold = 5.0;
    int span = 10;
    int numThreads = 4;
    tw_params optParam;
    // not in the reference: consumers that share one GPU (0: TW_CONSUMERS_PER_DEVICE, default 1) and the engine
    // batch of a consumer (0: 32).  The consumer count is min(numThreads, devices * perDevice).
    int consumersPerDevice = 0;
    int batch = 0;
    // bracket the level-0 window-average and polynomial-expansion launches of every consumer's engine with events
    // (tw_prof_select): the queue driver's roofline figures come from them (tools/bench_queue.cpp)
    bool profileKernels = false;
};

// What one consumer did, as seen by Manager::consumerStats() (a copy; the consumer thread owns the original).
struct ConsumerStats {
    int id = 0;
    int device = -1;          // id % deviceCount, -1 without a HIP device
    bool ready = false;       // the engine exists (or could not be created: engineError says why)
    std::string engineError;
    long pairs = 0;           // responses pushed since the last Manager::markEpoch()
    long batches = 0;         // engine batches submitted since then
    // event-bracketed level-0 launches since then: [0] window average + solve, [1] polynomial expansion
    double profMs[2] = {0, 0};
    int profLaunches[2] = {0, 0};
    // when the consumer worked since then (std::chrono::steady_clock, nanoseconds since its epoch): pop of its first
    // job, push of its last response, and the time it sat blocked on an EMPTY request queue in between — a driver
    // that knows its own clock derives each consumer's idle time from these (tools/bench_queue.cpp)
    long long firstJobNs = 0, lastResponseNs = 0;
    double waitMs = 0;
    // placement of the consumer thread (and, by inheritance / first touch, of its decode pool and of the page-locked
    // buffers its engine allocates): NUMA node of the GPU, -1 = not bound (TW_NUMA=0, no NUMA information, or none
    // of the node's CPUs is available to this process)
    std::string pciBusId;
    int numaNode = -1;
    std::vector<int> cpus;    // CPUs the consumer thread may run on after placement
};

// NUMA node and CPU list of a device from sysfs: /sys/bus/pci/devices/<bus id>/numa_node and
// /sys/devices/system/node/node<N>/cpulist (TW_SYSFS_ROOT replaces "/sys": tests).  false: unknown.
bool numa_cpus_of_device(int device, std::string* busid, int* node, std::vector<int>* cpus);
// Restrict the calling thread to `cpus` (those of them this process may use); false if none is usable.
bool bind_this_thread(const std::vector<int>& cpus);
std::vector<int> this_thread_cpus();

// Blocking MPMC queue, MessageQueue<T> of src/message_queue.h:50-118 (stop() wakes every waiter; as in the
// reference, items still queued at stop() are dropped — Appendix B#8 of SURVEY.md).
template <typename T>
class MessageQueue {
public:
    bool tryPop(T& out)
    {
        std::unique_lock<std::mutex> lk(m_);
        cv_.wait(lk, [&] { return !q_.empty() || !running_; });
        if (q_.empty() || !running_) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }
    // non-blocking: used by a consumer to fill a GPU batch with whatever is already queued
    bool tryPopNow(T& out)
    {
        std::lock_guard<std::mutex> lk(m_);
        if (q_.empty() || !running_) return false;
        out = std::move(q_.front());
        q_.pop_front();
        return true;
    }
    void push(T v)
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            q_.push_back(std::move(v));
        }
        cv_.notify_one();
    }
    void stop()
    {
        {
            std::lock_guard<std::mutex> lk(m_);
            running_ = false;
        }
        cv_.notify_all();
    }
    size_t size()
    {
        std::lock_guard<std::mutex> lk(m_);
        return q_.size();
    }

private:
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<T> q_;
    bool running_ = true;
};

// JPEG (baseline / progressive Huffman, 8-bit) -> luma plane, libjpeg's integer IDCT (jpeg_gray.cpp)
bool decode_jpeg_gray(const uint8_t* data, size_t n, std::vector<uint8_t>& img, int& w, int& h);
// 8-bit gray decode of a file: PGM (P5), PNG (zlib inflate + unfilter, libpng-1.5 gray conversion) and JPEG.
// Returns false if the file cannot be opened / decoded ("Can't open <path>", src/opticalflow.cpp:40,47).
bool load_gray(const std::string& path, std::vector<uint8_t>& img, int& w, int& h);
// load_gray, or — want_rows and an 8-bit non-interlaced non-palette PNG — the inflated but still filtered scanlines
// (ch 1-4: h rows of 1 + w * ch bytes for tw_submit_png8; ch 0: `img` is the gray image)
bool load_gray_or_png_rows(const std::string& path, bool want_rows, std::vector<uint8_t>& img, int& w, int& h, int& ch);
bool finish_png_rows_on_host(std::vector<uint8_t>& rows, int w, int h, int ch, std::vector<uint8_t>& gray);
// cv::resize(8-bit, INTER_LINEAR) — the "<= 5 px" reconcile of src/opticalflow.cpp:64-68.
void resize_u8_linear(const std::vector<uint8_t>& src, int sw, int sh, std::vector<uint8_t>& dst, int dw, int dh);

// Observer of src/observer.h:11-18
struct Observer {
    std::function<void(const Response&)> onNext;
    std::function<void(const std::string&)> onError;
    std::function<void(const Report&)> onCompleted;
};

// Consumer (src/consumer.{h,cpp}): one worker thread bound to one GPU; pulls requests, runs the engine, pushes
// responses.  Consumer i uses device i % deviceCount; with no HIP device every job answers ERROR (the engine
// has no CPU path).
// state shared between the Manager and its consumers
struct ConsumerShared {
    std::mutex m;
    std::condition_variable cv;
    std::vector<ConsumerStats> stats;  // [consumer id]
    std::atomic<int> epoch{0};
    std::atomic<long> inflight{0};  // pairs submitted to an engine and not yet collected, over all consumers
    bool profile = false;
};

class Consumer {
public:
    Consumer(int id, MessageQueue<Request>& req, MessageQueue<Response>& res, const tw_params& p, int batch,
             int decode_threads, int n_consumers = 1, ConsumerShared* shared = nullptr);
    ~Consumer();
    void start();
    void join();

private:
    void run();
    int id_;
    MessageQueue<Request>& req_;
    MessageQueue<Response>& res_;
    tw_params params_;
    int batch_;
    int decode_threads_;  // images of a batch are decoded by this many threads
    int n_consumers_;     // consumers on the same request queue: a consumer leaves the others their share of a short queue
    ConsumerShared* shared_;
    std::thread th_;
};

// Manager (src/manager.{h,cpp}): owns the queues and the consumers; a pump thread hands responses to the
// observer (the addon forwards them to the JS thread).
class Manager {
public:
    explicit Manager(Observer obs) : obs_(std::move(obs)) {}
    ~Manager();
    void start(const Parameter& p);
    int request(const std::string& expect_image, const std::string& target_image);
    // the same for an in-memory pair (see RawPair); the two names are only echoed in the response
    int requestRaw(const std::string& expect_name, const std::string& target_name, const RawPair& raw);
    int consumerCount() const { return (int)consumers_.size(); }
    // blocks until every consumer has bound its device and created its engine (or failed to)
    void waitReady();
    // snapshot of what every consumer did since the last markEpoch()
    std::vector<ConsumerStats> consumerStats();
    // every consumer zeroes its counters (and discards its pending kernel events) before the next job it takes;
    // call while the queue is empty and every response has arrived
    void markEpoch();
    PumpStats pumpStats();
    void stop();  // idempotent; completion is reported through onCompleted
    bool running() const { return running_; }

private:
    void work();
    Observer obs_;
    Parameter param_;
    MessageQueue<Request> requestQueue_;
    MessageQueue<Response> responseQueue_;
    std::vector<Consumer*> consumers_;
    std::thread pump_;
    std::atomic<bool> running_{false};
    std::atomic<bool> stopped_{false};
    Report report_;
    std::mutex report_m_;
    ConsumerShared shared_;
    PumpStats pump_stats_;
    double pump_sum_us_ = 0;
};

}  // namespace twhost
