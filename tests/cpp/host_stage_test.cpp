// CPU test of balm_amd/csrc/host_stage.h (the pinned-ring upload pipeline of the C ABI) against tests/cpp/fakehip: every byte
// arrives, every fill range is produced exactly once, ranges respect the unit, and no staging buffer is refilled while the
// (deliberately late) DMA out of it is still pending.  Built and run by tests/test_capi_cpu.py.
#include <cstdio>
#include <numeric>
#include "../../balm_amd/csrc/host_stage.h"

static int check(size_t bytes, size_t unit, int rounds) {
  std::vector<unsigned char> src(bytes), dst(bytes, 0);
  for (size_t i = 0; i < bytes; i++) src[i] = (unsigned char)((i * 2654435761u) >> 13);
  balm::PinnedRing ring;
  for (int r = 0; r < rounds; r++) {
    std::fill(dst.begin(), dst.end(), 0);
    std::vector<std::atomic<unsigned char>> seen((bytes + unit - 1) / unit);
    for (auto &s : seen) s.store(0);
    std::atomic<int> bad{0};
    hipError_t e = balm::staged_upload(ring, 0, nullptr, dst.data(), bytes, unit, [&](char *d, size_t off, size_t len) {
      if (off % unit || (len % unit && off + len != bytes)) bad++;
      for (size_t u = off / unit; u < (off + len + unit - 1) / unit; u++)
        if (seen[u].fetch_add(1) != 0) bad++;
      std::memcpy(d, src.data() + off, len);
    });
    if (e != hipSuccess) { printf("FAIL upload rc=%d\n", e); return 1; }
    hipStreamSynchronize(nullptr);
    for (auto &s : seen) if (s.load() != 1) bad++;
    if (bad.load() || std::memcmp(src.data(), dst.data(), bytes)) {
      printf("FAIL bytes=%zu unit=%zu round=%d bad=%d\n", bytes, unit, r, bad.load());
      return 1;
    }
  }
  ring.release();
  return 0;
}

int main() {
  int rc = 0;
  rc |= check(1000, 8, 2);                                   // the small path
  rc |= check(((size_t)1 << 20) + 64, 64, 2);                // just above it: one chunk
  rc |= check((size_t)5 * (16 << 20) + 12345 * 12, 12, 2);   // 5+ chunks, 12-byte units (points), ring reused across calls
  rc |= check((size_t)3001 * 177 * 80, (size_t)177 * 80, 2); // cluster-table rows of a 177-pose window
  rc |= check((size_t)70 * 1024 * 80 * 9, (size_t)1024 * 80, 1);   // the widest window's rows (80 KiB units)
  // parallel_ranges covers [0, n) exactly once
  std::vector<std::atomic<int>> hit(100003);
  for (auto &h : hit) h.store(0);
  balm::parallel_ranges(hit.size(), 1000, [&](size_t lo, size_t hi) { for (size_t i = lo; i < hi; i++) hit[i]++; });
  for (auto &h : hit) if (h.load() != 1) { printf("FAIL parallel_ranges\n"); rc = 1; break; }
  if (!rc) printf("host_stage ok (%d pool threads)\n", balm::HostPool::get().workers());
  return rc;
}
