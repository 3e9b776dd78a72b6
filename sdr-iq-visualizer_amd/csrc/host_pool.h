// host_pool.h — a few helper threads that move bytes between the caller's pageable arrays and the
// library's pinned staging buffers (the numpy boundary of sdrk_exec_host).  One memcpy thread
// sustains 10-15 GB/s; PCIe Gen5 x16 moves ~55 GB/s each way, so the staging copies, not the
// link, bound a single-threaded host path.  The pool is process-wide and shared by every plan:
// a copy is cut into 1 MiB pieces on one queue, the calling thread works on the queue too (so
// a pool of zero helpers degrades to a plain memcpy), and callers on different threads (one per
// GPU in sharding.py) interleave their pieces instead of serialising.
#pragma once
#include <atomic>
#include <condition_variable>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <thread>
#include <vector>

namespace sdrk {

class CopyPool {
public:
    static CopyPool& get() {
        static CopyPool pool;
        return pool;
    }

    int helpers() const { return n_helpers_; }

    // dst[0..bytes) = src[0..bytes); returns when every byte has been copied.
    void copy(void* dst, const void* src, size_t bytes) {
        if (bytes == 0) return;
        if (bytes < 2 * PIECE || n_helpers_ == 0) {
            memcpy(dst, src, bytes);
            return;
        }
        start_helpers();
        std::atomic<size_t> remaining{(bytes + PIECE - 1) / PIECE};
        {
            std::lock_guard<std::mutex> g(m_);
            for (size_t off = 0; off < bytes; off += PIECE)
                q_.push_back(Piece{static_cast<char*>(dst) + off, static_cast<const char*>(src) + off,
                                   bytes - off < PIECE ? bytes - off : PIECE, &remaining});
        }
        cv_.notify_all();
        Piece p;
        while (pop(p)) run(p);                       // may also run pieces of other callers
        while (remaining.load(std::memory_order_acquire) != 0) std::this_thread::yield();
    }

    CopyPool(const CopyPool&) = delete;
    CopyPool& operator=(const CopyPool&) = delete;

private:
    static constexpr size_t PIECE = 1u << 20;
    struct Piece {
        char* dst;
        const char* src;
        size_t n;
        std::atomic<size_t>* remaining;
    };

    CopyPool() {
        // SDRK_HOST_THREADS = number of helper threads (0 = copy on the calling thread only).
        // Default: half the CPUs this process may run on, at most 7 helpers.
        int n = -1;
        if (const char* env = getenv("SDRK_HOST_THREADS")) n = atoi(env);
        if (n < 0) {
            unsigned hw = std::thread::hardware_concurrency();
            cpu_set_t set;
            if (sched_getaffinity(0, sizeof set, &set) == 0) hw = (unsigned)CPU_COUNT(&set);
            n = (int)(hw / 2);
            if (n > 7) n = 7;
        }
        if (n > 64) n = 64;
        n_helpers_ = n;
    }

    ~CopyPool() {
        {
            std::lock_guard<std::mutex> g(m_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : threads_) t.join();
    }

    void start_helpers() {
        if (started_.load(std::memory_order_acquire)) return;
        std::lock_guard<std::mutex> g(m_);
        if (started_.load(std::memory_order_relaxed)) return;
        for (int i = 0; i < n_helpers_; ++i) threads_.emplace_back([this] { worker(); });
        started_.store(true, std::memory_order_release);
    }

    bool pop(Piece& p) {
        std::lock_guard<std::mutex> g(m_);
        if (q_.empty()) return false;
        p = q_.front();
        q_.pop_front();
        return true;
    }

    static void run(const Piece& p) {
        memcpy(p.dst, p.src, p.n);
        p.remaining->fetch_sub(1, std::memory_order_release);
    }

    void worker() {
        for (;;) {
            Piece p;
            {
                std::unique_lock<std::mutex> lk(m_);
                cv_.wait(lk, [this] { return stop_ || !q_.empty(); });
                if (stop_ && q_.empty()) return;
                p = q_.front();
                q_.pop_front();
            }
            run(p);
        }
    }

    int n_helpers_ = 0;
    std::atomic<bool> started_{false};
    std::mutex m_;
    std::condition_variable cv_;
    std::deque<Piece> q_;
    std::vector<std::thread> threads_;
    bool stop_ = false;
};

}  // namespace sdrk
