// eb_pieces.h -- host threads that fill page-locked pieces from an input (the eBWT file, the document-array file, or a caller's
// pageable buffer) while earlier pieces are uploaded: shared by the index loader (eb_index.hip) and the document-array loader
// (eb_navigate.hip).  Internal to the library.
#pragma once
#include <stdint.h>
#include <string.h>
#include <unistd.h>

#include <condition_variable>
#include <mutex>
#include <thread>
#include <vector>

// Host threads that fill page-locked pieces from the input (pread of the file: page cache -> piece with no page faults of a
// mapping in between; or memcpy from the caller's buffer).  They live for one index creation and take 4 MB chunks off a queue, so
// that several pieces are being filled while the one before them is uploaded and packed (round 4 started and joined a set of
// threads per piece and read one piece at a time: 0.20 s of reads for the 7.55 GB file, on the critical path, plus ~2000 thread
// starts).
struct PieceReaders {
    struct Chunk { uint8_t* dst; uint64_t off, len; int piece; };
    const uint8_t* src = nullptr; // host buffer, or null: read fd
    int fd = -1;
    std::mutex mu;
    std::condition_variable cv_work, cv_done;
    std::vector<Chunk> queue;
    size_t head = 0;
    std::vector<int> left;        // chunks still to be read, per piece
    bool stop = false, io_error = false;
    std::vector<std::thread> pool;
    void start(unsigned threads, size_t n_pieces) {
        left.assign(n_pieces, 0);
        for (unsigned t = 0; t < threads; ++t) pool.emplace_back([this] { run(); });
    }
    void run() {
        for (;;) {
            Chunk c;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv_work.wait(lk, [&] { return stop || head < queue.size(); });
                if (head >= queue.size()) return; // stop, nothing left
                c = queue[head++];
            }
            bool ok = true;
            if (src) memcpy(c.dst, src + c.off, c.len);
            else
                for (uint64_t got = 0; got < c.len;) {
                    ssize_t r = pread(fd, c.dst + got, c.len - got, (off_t)(c.off + got));
                    if (r <= 0) { ok = false; break; }
                    got += (uint64_t)r;
                }
            std::lock_guard<std::mutex> lk(mu);
            if (!ok) io_error = true;
            if (--left[(size_t)c.piece] == 0) cv_done.notify_all();
        }
    }
    void post(int piece, uint8_t* dst, uint64_t off, uint64_t len) {
        const uint64_t CH = 4ull << 20;
        std::lock_guard<std::mutex> lk(mu);
        int k = 0;
        for (uint64_t a = 0; a < len; a += CH, ++k) queue.push_back({dst + a, off + a, len - a < CH ? len - a : CH, piece});
        left[(size_t)piece] = k;
        cv_work.notify_all();
    }
    bool wait(int piece) { // false: read error
        std::unique_lock<std::mutex> lk(mu);
        cv_done.wait(lk, [&] { return left[(size_t)piece] == 0; });
        return !io_error;
    }
    ~PieceReaders() {
        { std::lock_guard<std::mutex> lk(mu); stop = true; head = queue.size(); }
        cv_work.notify_all();
        for (std::thread& t : pool) t.join();
    }
};
