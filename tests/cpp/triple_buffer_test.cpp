// Terminating, assertion-bearing counterpart of the reference's soak tests
// (reference test/triple_buffer_test.cpp:17-101, which only stop on SIGINT):
// one producer, one consumer, checks that the consumer always sees a complete,
// never-older frame and that the final frame is delivered.
#include <array>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <thread>

#include "irmv_detection/triple_buffer.hpp"

struct Frame { long id = -1; long payload[64] = {0}; };

int main()
{
  std::array<Frame, 3> buffers;
  irmv_detection::TripleBuffer<Frame> tb(buffers);
  constexpr long kFrames = 200000;
  std::atomic<bool> torn{false};
  long last_seen = -1, consumed = 0;
  std::thread producer([&] {
    for (long i = 0; i < kFrames; i++) {
      Frame * f = tb.get_producer_buffer();
      f->id = i;
      for (long & p : f->payload) p = i;
      tb.producer_commit();
    }
  });
  while (last_seen != kFrames - 1) {
    Frame * f = tb.get_consumer_buffer();
    for (long p : f->payload)
      if (p != f->id) torn = true;
    if (f->id <= last_seen) torn = true;  // must be strictly newer
    last_seen = f->id;
    consumed++;
  }
  producer.join();
  const bool none_pending = tb.try_get_consumer_buffer() == nullptr;
  std::printf("consumed %ld of %ld, last %ld, torn %d, none_pending %d\n", consumed, kFrames, last_seen, int(torn.load()), int(none_pending));
  return (!torn && last_seen == kFrames - 1 && none_pending && consumed >= 1) ? 0 : 1;
}
