// Host-only stress of csrc/lane_format.hip under AddressSanitizer / UBSan (tools/asan_lane_format.sh):
// random well-formed record streams through the three formatting calls with exactly sized
// buffers, then malformed headers (negative counts, offsets and extents outside the buffer,
// overflow bits), which must be refused without a read outside `packed`.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>

#include "ganet_hip.h"

static std::mt19937_64 rng(12345);
static int rnd(int lo, int hi) { return lo + (int)(rng() % (uint64_t)(hi - lo + 1)); }

static std::vector<int32_t> make_packed(int B, int by_center, bool wild) {
  std::vector<int32_t> head((size_t)B * 8, 0), rec;
  for (int b = 0; b < B; ++b) {
    const int ng = rnd(0, 6), nk = rnd(0, 60), nc = by_center ? rnd(ng, ng + 4) : 0;
    int32_t *h = &head[(size_t)b * 8];
    h[0] = nk; h[1] = nc; h[2] = ng; h[3] = 0; h[4] = (int32_t)rec.size();
    const int ngc = by_center ? ng : 0;
    h[5] = 7 * nk + 3 * nc + 2 * ngc + ng;
    auto putf = [&](float f) { int32_t v; std::memcpy(&v, &f, 4); rec.push_back(v); };
    for (int i = 0; i < 2 * nk; ++i) rec.push_back(rnd(0, 200));                 // kpt_xy
    for (int i = 0; i < 2 * nk; ++i) putf(wild && rnd(0, 50) == 0 ? (rnd(0, 1) ? NAN : 3e38f)
                                                                  : (float)rnd(-50, 2000) * 0.37f);
    for (int i = 0; i < 2 * nk; ++i) putf((float)rnd(-50, 2000) * 0.41f);       // kpt_root
    for (int i = 0; i < nk; ++i) rec.push_back(ng ? rnd(-1, ng - 1) : -1);      // kpt_gid
    for (int i = 0; i < 3 * nc; ++i) rec.push_back(rnd(0, 300));                // centres + gid
    for (int i = 0; i < 2 * ngc; ++i) rec.push_back(rnd(0, 300));               // group_center
    for (int i = 0; i < ng; ++i) rec.push_back(rnd(0, 60));                     // group_count
  }
  head.insert(head.end(), rec.begin(), rec.end());
  return head;
}

static int run(const std::vector<int32_t> &packed, int B, int by_center, int threads, long long words) {
  ganet_lane_format f;
  std::memset(&f, 0, sizeof f);
  f.B = B; f.by_center = by_center; f.downscale = 8; f.points_thr = rnd(0, 4);
  f.float32_scalars = rnd(0, 1); f.threads = threads; f.img_h = 320; f.img_w = 800;
  f.crop_x0 = 0; f.crop_y0 = 270; f.crop_x1 = 1640; f.crop_y1 = 590;
  f.packed_words = words;
  size_t need = 0;
  int rc = ganet_lanes_from_packed(packed.data(), &f, nullptr, 0, &need);
  if (rc != GANET_ERR_WORKSPACE) return rc;
  std::vector<int32_t> lanes(need);                      // exactly sized: ASan sees any overrun
  rc = ganet_lanes_from_packed(packed.data(), &f, lanes.data(), need, &need);
  if (rc != GANET_OK) return rc;
  std::vector<size_t> ts((size_t)2 * B + 1);
  rc = ganet_culane_text(lanes.data(), B, threads, nullptr, 0, ts.data());
  if (rc != GANET_ERR_WORKSPACE && !(rc == GANET_OK && ts[B] == 0)) return 1000 + rc;
  std::vector<char> text(ts[B] ? ts[B] : 1);
  rc = ganet_culane_text(lanes.data(), B, threads, text.data(), ts[B], ts.data());
  if (rc != GANET_OK) return 2000 + rc;
  std::vector<int32_t> hs(56), ls((size_t)B + 1);
  for (int i = 0; i < 56; ++i) hs[i] = 160 + 10 * i;
  int nl = 0;
  for (int b = 0; b < B; ++b) nl += lanes[(size_t)b * 4];
  std::vector<int32_t> xs((size_t)nl * 56 + 1);
  rc = ganet_tusimple_from_lanes(lanes.data(), B, hs.data(), 56, -2, threads, xs.data(), (size_t)nl * 56, ls.data());
  if (rc != GANET_OK && rc != GANET_ERR_NOT_FINITE) return 3000 + rc;
  return GANET_OK;
}

int main() {
  int ok = 0, refused = 0, notfinite = 0;
  for (int it = 0; it < 3000; ++it) {
    const int B = rnd(1, 40), by_center = rnd(0, 1), threads = rnd(0, 5);
    const bool wild = it % 3 == 0;
    std::vector<int32_t> packed = make_packed(B, by_center, wild);
    long long words = (long long)packed.size();
    const int mode = it % 7;
    if (mode == 3) packed[(size_t)rnd(0, B - 1) * 8 + rnd(0, 2)] = -rnd(1, 1 << 30);          // negative count
    if (mode == 4) packed[(size_t)rnd(0, B - 1) * 8 + 4] = rnd(1 << 20, 1 << 30);              // offset far outside
    if (mode == 5) words = rnd(0, (int)packed.size() - 1);                                      // buffer cut short
    if (mode == 6) packed[(size_t)rnd(0, B - 1) * 8 + 3] = rnd(1, 15);                          // overflow bits
    {   // the buffer the library may read is exactly `words` long
      std::vector<int32_t> exact(packed.begin(), packed.begin() + (words < (long long)packed.size() && words >= 8LL * B ? words : (long long)packed.size()));
      const int rc = run(exact, B, by_center, threads, words);
      if (rc == GANET_OK) ++ok;
      else if (rc == GANET_ERR_BAD_SHAPE) ++refused;
      else if (rc == GANET_ERR_NOT_FINITE) ++notfinite;
      else { std::printf("unexpected status %d at iteration %d (mode %d)\n", rc, it, mode); return 1; }
      if (mode >= 4 && mode <= 6 && rc == GANET_OK && mode != 5) { std::printf("malformed stream accepted (mode %d)\n", mode); return 1; }
    }
  }
  std::printf("lane_format fuzz: %d formatted, %d refused, %d not finite\n", ok, refused, notfinite);
  return 0;
}
