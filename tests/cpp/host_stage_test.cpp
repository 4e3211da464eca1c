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

// strided point containers (48-byte PointXYZINormal-like elements, ragged and empty scans) -> packed xyz / xyz+aux records
struct Pt48 { float x, y, z, pad; float n[4]; float intensity, curvature, c2, c3; };
static int check_points(const std::vector<long> &counts, bool aux, size_t stride) {
  std::vector<std::vector<char>> clouds(counts.size());
  std::vector<const void *> base(counts.size());
  std::vector<float> want;
  unsigned v = 12345;
  auto next = [&] { v = v * 1664525u + 1013904223u; return (float)(v >> 8) * (1.0f / 65536.0f); };
  for (size_t k = 0; k < counts.size(); k++) {
    clouds[k].assign((size_t)counts[k] * stride + 16, (char)0x5a);
    base[k] = counts[k] ? clouds[k].data() : nullptr;
    for (long i = 0; i < counts[k]; i++) {
      float *p = reinterpret_cast<float *>(clouds[k].data() + (size_t)i * stride);
      for (int c = 0; c < 3; c++) { p[c] = next(); want.push_back(p[c]); }
      if (aux) { float *w = reinterpret_cast<float *>(clouds[k].data() + (size_t)i * stride + 32); *w = (float)(i % 7); want.push_back(*w); }
    }
  }
  balm::StridedPoints sp;
  if (!sp.set((int)counts.size(), base.data(), counts.data(), stride, aux ? 32 : balm::StridedPoints::NO_AUX)) { printf("FAIL StridedPoints::set\n"); return 1; }
  std::vector<float> got(want.size() + 4, -1.0f);
  balm::PinnedRing ring;
  hipError_t e = balm::staged_points(ring, 0, nullptr, got.data(), sp);
  hipStreamSynchronize(nullptr);
  ring.release();
  if (e != hipSuccess || (size_t)sp.total() * sp.rec() != want.size() * 4 || std::memcmp(got.data(), want.data(), want.size() * 4) || got[want.size()] != -1.0f) {
    printf("FAIL points scans=%zu aux=%d stride=%zu\n", counts.size(), (int)aux, stride);
    return 1;
  }
  // any sub-range, from any (misaligned) destination
  std::vector<float> part(3000 * 4 + 8);
  const long n = sp.total();
  for (long p0 : {0l, 1l, 5l, n / 2, n - 7}) {
    if (p0 < 0 || p0 >= n) continue;
    const long np = std::min<long>(3000, n - p0);
    for (int mis = 0; mis < 4; mis++) {
      sp.gather(reinterpret_cast<char *>(part.data() + mis), p0, np);
      if (std::memcmp(part.data() + mis, want.data() + (size_t)p0 * (sp.rec() / 4), (size_t)np * sp.rec())) { printf("FAIL gather p0=%ld mis=%d\n", p0, mis); return 1; }
    }
  }
  return 0;
}

int main() {
  int rc = 0;
  rc |= check_points({5, 0, 1, 100003, 0, 7, 250001, 3}, false, 48);
  rc |= check_points({5, 0, 1, 100003, 0, 7, 250001, 3}, true, 48);
  rc |= check_points({1000, 999, 40001}, false, 12);          // already packed
  rc |= check_points({1000, 999, 40001}, false, 16);
  rc |= check_points({17}, false, 48);
  { balm::StridedPoints bad; const long c1[1] = {4}; const void *b1[1] = {nullptr};
    if (bad.set(1, b1, c1, 48) || bad.set(0, nullptr, nullptr, 10)) { printf("FAIL StridedPoints accepted a bad container\n"); rc = 1; } }
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
  // core_groups: the pool threads' CPU groups partition the set's cores -- pairwise disjoint, inside the set, none empty, together the whole set
  {
    cpu_set_t all;
    sched_getaffinity(0, sizeof(all), &all);
    for (int n : {1, 2, 3, 15}) {
      const std::vector<cpu_set_t> g = balm::core_groups(all, n);
      if ((int)g.size() != n) { printf("FAIL core_groups: %zu groups for %d threads\n", g.size(), n); rc = 1; continue; }
      cpu_set_t uni;
      CPU_ZERO(&uni);
      bool whole = true, ok = true;                      // (fewer cores than threads: every group is the whole set)
      for (auto &q : g) whole = whole && CPU_EQUAL(&q, &all);
      for (int a = 0; a < n && !whole; a++) {
        cpu_set_t in;
        CPU_AND(&in, &g[(size_t)a], &all);
        if (CPU_COUNT(&g[(size_t)a]) == 0 || !CPU_EQUAL(&in, &g[(size_t)a])) ok = false;
        for (int b = a + 1; b < n; b++) { cpu_set_t x; CPU_AND(&x, &g[(size_t)a], &g[(size_t)b]); if (CPU_COUNT(&x)) ok = false; }
        CPU_OR(&uni, &uni, &g[(size_t)a]);
      }
      if (!whole && (!ok || !CPU_EQUAL(&uni, &all))) { printf("FAIL core_groups(%d): not a partition of the process's CPUs\n", n); rc = 1; }
    }
  }
  if (!rc) printf("host_stage ok (%d pool threads)\n", balm::HostPool::get().workers());
  return rc;
}
