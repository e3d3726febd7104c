// Lock-free three-slot hand-off between one producer (camera) and one consumer
// (detector) -- API-compatible with the reference's
// include/irmv_detection/triple_buffer.hpp:15-49 (get_producer_buffer,
// producer_commit, get_consumer_buffer).  The slots themselves are the engine's
// pinned host frame buffers; the consumer then starts the async H2D + graph step.
//
// Design: a single atomic word holds the index of the "middle" slot plus a fresh
// bit.  The producer publishes by exchanging its slot with the middle one and
// setting the bit; the consumer takes the middle slot only when the bit is set,
// clearing it in the same atomic exchange -- so, unlike the reference
// (exchange, then a separate store(false): :36-38), a commit that lands between
// the two consumer steps cannot lose its wake-up.
#pragma once

#include <array>
#include <atomic>
#include <cstdint>

namespace irmv_detection
{
template <typename Buffer>
class TripleBuffer
{
public:
  explicit TripleBuffer(std::array<Buffer, 3> & buffers) : slots_{&buffers[0], &buffers[1], &buffers[2]} {}

  Buffer * get_producer_buffer() { return slots_[write_]; }

  // Make the slot just written the newest complete one; never blocks.
  void producer_commit()
  {
    const uint32_t prev = middle_.exchange(write_ | kFresh, std::memory_order_acq_rel);
    write_ = prev & kIndex;
    middle_.notify_one();
  }

  // Block until a frame newer than the last one consumed exists, then return it.
  Buffer * get_consumer_buffer()
  {
    for (;;) {
      uint32_t cur = middle_.load(std::memory_order_acquire);
      if (cur & kFresh) {
        if (middle_.compare_exchange_weak(cur, read_, std::memory_order_acq_rel)) {
          read_ = cur & kIndex;
          return slots_[read_];
        }
      } else {
        middle_.wait(cur, std::memory_order_acquire);
      }
    }
  }

  // Non-blocking variant: nullptr if nothing new.
  Buffer * try_get_consumer_buffer()
  {
    uint32_t cur = middle_.load(std::memory_order_acquire);
    while (cur & kFresh) {
      if (middle_.compare_exchange_weak(cur, read_, std::memory_order_acq_rel)) {
        read_ = cur & kIndex;
        return slots_[read_];
      }
    }
    return nullptr;
  }

private:
  static constexpr uint32_t kFresh = 4u, kIndex = 3u;
  std::array<Buffer *, 3> slots_;
  uint32_t write_ = 0;               // producer-owned
  uint32_t read_ = 2;                // consumer-owned
  std::atomic<uint32_t> middle_{1};  // shared: index | fresh bit
};
}  // namespace irmv_detection
