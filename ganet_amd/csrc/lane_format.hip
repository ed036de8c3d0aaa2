// Host side of the path after grouping: decoded + grouped records -> the reference's final
// result formats, for a whole batch per call and without per-point interpreter work.
//
//   ganet_lanes_from_packed   PostProcessor.lane_post_process (tools/ganet/post_process.py:216-242)
//                             + adjust_result (tools/ganet/tusimple/test_dataset.py:49-83 ==
//                             tools/ganet/culane/test_dataset.py:79-113) on the record stream
//                             of ganet_decode_pack
//   ganet_tusimple_from_lanes tusimple_convert_formal / get_line_intersection
//                             (tools/ganet/common.py:118-147)
//   ganet_culane_text         out_result (tools/ganet/culane/test_dataset.py:138-148)
//
// Plain C++ (no device code); compiled with -ffp-contract=off: the reference evaluates
// `int(v * ratio + offset)` as two rounded operations, a fused multiply-add would differ in
// the last bit and with it the truncated integer.

#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <thread>
#include <vector>

#include "common.h"

namespace {

struct ImageView {
  const int32_t *kpt_align, *kpt_root, *kpt_gid, *group_center;
  int nk, nc, ng, ngc;
};

// record b of ganet_decode_pack (include/ganet_hip.h); false if the record is unusable
bool view_image(const int32_t *packed, long long packed_words, int B, int b, int by_center,
                ImageView *v) {
  const int32_t *h = packed + (size_t)b * 8;
  if (h[3] & 15) return false;   // capacity overflow / record not packed
  if (h[0] < 0 || h[1] < 0 || h[2] < 0 || h[4] < 0 || h[5] < 0) return false;
  const int32_t *r = packed + (size_t)B * 8 + h[4];
  v->nk = h[0];
  v->nc = h[1];
  v->ng = h[2];
  v->ngc = by_center ? v->ng : 0;
  if ((long long)h[5] != 7ll * v->nk + 3ll * v->nc + 2ll * v->ngc + v->ng) return false;
  // the record has to lie inside the caller's buffer
  if (8ll * B + h[4] + h[5] > packed_words) return false;
  v->kpt_align = r + 2 * (size_t)v->nk;
  v->kpt_root = r + 4 * (size_t)v->nk;
  v->kpt_gid = r + 6 * (size_t)v->nk;
  v->group_center = r + 7 * (size_t)v->nk + 3 * (size_t)v->nc;
  return true;
}

inline float as_f32(int32_t bits) {
  float f;
  std::memcpy(&f, &bits, 4);
  return f;
}

// Python's int(): truncation towards zero; NaN / infinity raise there, and so does a value
// no int32 holds here
inline bool to_int(double t, int32_t *out) {
  if (!(t > -2147483649.0 && t < 2147483648.0)) return false;   // NaN fails both
  *out = (int32_t)t;   // the conversion truncates towards zero
  return true;
}

struct Scale {
  double ratio, offset;
  int f32;   // NumPy >= 2 scalar rules: float32 * python float stays float32
};

// int(align * downscale * ratio + offset) with align an np.float32
inline bool point_coord(float a, int downscale, const Scale &s, int32_t *out) {
  const float v = a * (float)downscale;                       // float32 array * python int
  if (s.f32) {
    const float t = v * (float)s.ratio;
    return to_int((double)(t + (float)s.offset), out);
  }
  const double t = (double)v * s.ratio;
  return to_int(t + s.offset, out);
}

// root is widened to float64 before the scaling (ganet_head.py:236)
inline bool center_coord(float r, int downscale, const Scale &s, int32_t *out) {
  const double v = (double)r * (double)downscale;
  const double t = v * s.ratio;
  return to_int(t + s.offset, out);
}

inline bool cluster_coord(int32_t c, int downscale, const Scale &s, int32_t *out) {
  const double v = (double)((long long)c * downscale);        // np.int64 * python int
  const double t = v * s.ratio;
  return to_int(t + s.offset, out);
}

// threads are started per call, which costs more than formatting a few thousand points:
// `work` (points of the whole batch) below WORK_PER_THREAD a thread keeps the call serial
constexpr long long WORK_PER_THREAD = 16384;

// An exception inside f (a vector that cannot grow) must neither leave a worker thread - that
// is std::terminate - nor unwind the caller past threads that are still joinable: every body
// catches, every thread is joined, and the failure is thrown again on the calling thread, where
// guarded() turns it into GANET_ERR_WORKSPACE like any other allocation failure.
template <typename F>
void for_images(int B, int threads, long long work, F f) {
  if ((long long)threads > work / WORK_PER_THREAD) threads = (int)(work / WORK_PER_THREAD);
  if (threads <= 1 || B < 2 * threads) {
    for (int b = 0; b < B; ++b) f(b);
    return;
  }
  std::atomic<bool> failed(false);
  auto stripe = [&](int t) {
    try {
      for (int b = t; b < B && !failed.load(std::memory_order_relaxed); b += threads) f(b);
    } catch (...) {
      failed.store(true, std::memory_order_relaxed);
    }
  };
  std::vector<std::thread> pool;
  int started = 0;
  try {
    pool.reserve(threads);
    for (; started < threads; ++started) pool.emplace_back(stripe, started);
  } catch (...) {   // no more threads to be had: the calling thread takes the rest
  }
  for (int t = started; t < threads; ++t) stripe(t);
  for (auto &th : pool) th.join();
  if (failed.load()) throw std::bad_alloc();
}

// decimal text of an int followed by ".00": what "{:.2f}".format(int) prints
inline char *put_int_2f(char *p, int32_t v) {
  if (v >= 0 && v < 10000) {   // pixel coordinates: at most four digits
    if (v >= 1000) *p++ = (char)('0' + v / 1000);
    if (v >= 100) *p++ = (char)('0' + v / 100 % 10);
    if (v >= 10) *p++ = (char)('0' + v / 10 % 10);
    *p++ = (char)('0' + v % 10);
    *p++ = '.';
    *p++ = '0';
    *p++ = '0';
    return p;
  }
  char tmp[16];
  int n = 0;
  long long a = v;
  if (a < 0) {
    *p++ = '-';
    a = -a;
  }
  do {
    tmp[n++] = (char)('0' + a % 10);
    a /= 10;
  } while (a);
  while (n) *p++ = tmp[--n];
  *p++ = '.';
  *p++ = '0';
  *p++ = '0';
  return p;
}

// no C++ exception may cross the C boundary: out of memory reads as a workspace error
template <typename F>
int guarded(F body) {
  try {
    return body();
  } catch (...) {
    return GANET_ERR_WORKSPACE;
  }
}

int lanes_from_packed(const int32_t *packed, const ganet_lane_format *f, int32_t *lanes,
                      size_t lanes_words, size_t *needed_words) {
  if (!packed || !f || !needed_words) return GANET_ERR_NULL;
  const int B = f->B;
  if (B <= 0 || f->img_h <= 0 || f->img_w <= 0 || f->downscale <= 0) return GANET_ERR_BAD_SHAPE;
  if (f->packed_words < 8ll * B) return GANET_ERR_BAD_SHAPE;
  const Scale sx = {(f->crop_x1 - f->crop_x0) / (double)f->img_w, f->crop_x0, f->float32_scalars};
  const Scale sy = {(f->crop_y1 - f->crop_y0) / (double)f->img_h, f->crop_y0, f->float32_scalars};

  long long work = 0;
  for (int b = 0; b < B; ++b) work += packed[(size_t)b * 8] > 0 ? packed[(size_t)b * 8] : 0;
  // pass 1: members per group -> lanes kept and words per image
  std::vector<int> words(B, 0), nlanes(B, 0), bad(B, 0);
  for_images(B, f->threads, work, [&](int b) {
    ImageView v;
    if (!view_image(packed, f->packed_words, B, b, f->by_center, &v)) {
      bad[b] = 1;
      return;
    }
    std::vector<int> cnt(v.ng > 0 ? v.ng : 0, 0);
    for (int k = 0; k < v.nk; ++k) {
      const int g = v.kpt_gid[k];
      if (g >= 0 && g < v.ng) ++cnt[g];
    }
    int w = 2 * v.ngc, nl = 0;
    for (int g = 0; g < v.ng; ++g)
      if (cnt[g] > 1 && cnt[g] > f->points_thr) {
        w += 2 + 4 * cnt[g];
        ++nl;
      }
    words[b] = w;
    nlanes[b] = nl;
  });
  size_t total = (size_t)B * 4;
  for (int b = 0; b < B; ++b) {
    if (bad[b]) return GANET_ERR_BAD_SHAPE;
    total += (size_t)words[b];
  }
  *needed_words = total;
  if (!lanes || lanes_words < total) return GANET_ERR_WORKSPACE;
  size_t off = 0;
  for (int b = 0; b < B; ++b) {
    int32_t *h = lanes + (size_t)b * 4;
    ImageView v;
    view_image(packed, f->packed_words, B, b, f->by_center, &v);
    h[0] = nlanes[b];
    h[1] = v.ngc;
    h[2] = (int32_t)off;
    h[3] = words[b];
    off += (size_t)words[b];
  }

  // pass 2: the lanes in group order, their points in keypoint order
  for_images(B, f->threads, work, [&](int b) {
    ImageView v;
    view_image(packed, f->packed_words, B, b, f->by_center, &v);
    int32_t *rec = lanes + (size_t)B * 4 + lanes[(size_t)b * 4 + 2];
    std::vector<int> cnt(v.ng > 0 ? v.ng : 0, 0), start(v.ng > 0 ? v.ng : 0, -1);
    for (int k = 0; k < v.nk; ++k) {
      const int g = v.kpt_gid[k];
      if (g >= 0 && g < v.ng) ++cnt[g];
    }
    int pos = 0;
    for (int g = 0; g < v.ng; ++g)
      if (cnt[g] > 1 && cnt[g] > f->points_thr) {
        start[g] = pos;
        rec[pos] = g;            // id_class: the group's index, gaps allowed
        rec[pos + 1] = cnt[g];
        pos += 2 + 4 * cnt[g];
      }
    std::vector<int> fill(v.ng > 0 ? v.ng : 0, 0);
    bool ok = true;
    for (int k = 0; k < v.nk; ++k) {
      const int g = v.kpt_gid[k];
      if (g < 0 || g >= v.ng || start[g] < 0) continue;
      const int j = fill[g]++;
      int32_t *pts = rec + start[g] + 2 + 2 * j;
      int32_t *cts = rec + start[g] + 2 + 2 * cnt[g] + 2 * j;
      ok &= point_coord(as_f32(v.kpt_align[2 * k]), f->downscale, sx, &pts[0]);
      ok &= point_coord(as_f32(v.kpt_align[2 * k + 1]), f->downscale, sy, &pts[1]);
      ok &= center_coord(as_f32(v.kpt_root[2 * k]), f->downscale, sx, &cts[0]);
      ok &= center_coord(as_f32(v.kpt_root[2 * k + 1]), f->downscale, sy, &cts[1]);
    }
    for (int g = 0; g < v.ngc; ++g) {
      ok &= cluster_coord(v.group_center[2 * g], f->downscale, sx, &rec[pos + 2 * g]);
      ok &= cluster_coord(v.group_center[2 * g + 1], f->downscale, sy, &rec[pos + 2 * g + 1]);
    }
    if (!ok) bad[b] = 1;
  });
  for (int b = 0; b < B; ++b)
    if (bad[b]) return GANET_ERR_NOT_FINITE;
  return GANET_OK;
}

int tusimple_from_lanes(const int32_t *lanes, int B, const int32_t *h_samples, int n_h, int reg_x,
                        int threads, int32_t *xs, size_t xs_words, int32_t *lane_start) {
  if (!lanes || !h_samples || !xs || !lane_start) return GANET_ERR_NULL;
  if (B <= 0 || n_h < 0) return GANET_ERR_BAD_SHAPE;
  lane_start[0] = 0;
  for (int b = 0; b < B; ++b) lane_start[b + 1] = lane_start[b] + lanes[(size_t)b * 4];
  if (xs_words < (size_t)lane_start[B] * (size_t)n_h) return GANET_ERR_WORKSPACE;
  std::vector<int> bad(B, 0);
  for_images(B, threads, (long long)lane_start[B] * n_h * 4, [&](int b) {
    const int32_t *rec = lanes + (size_t)B * 4 + lanes[(size_t)b * 4 + 2];
    int pos = 0;
    for (int l = 0; l < lanes[(size_t)b * 4]; ++l) {
      const int n = rec[pos + 1];
      const int32_t *pts = rec + pos + 2;
      int32_t *row = xs + ((size_t)lane_start[b] + l) * n_h;
      for (int s = 0; s < n_h; ++s) {
        const int y = h_samples[s];
        int x = reg_x;
        for (int i = 0; i + 1 < n; ++i) {
          const int x0 = pts[2 * i], y0 = pts[2 * i + 1], x1 = pts[2 * i + 2], y1 = pts[2 * i + 3];
          const int lo = y0 < y1 ? y0 : y1, hi = y0 < y1 ? y1 : y0;
          if (lo <= y && y <= hi && lo != hi) {
            // Python integers: the differences are formed without wrap-around
            const double k = ((double)x1 - (double)x0) / ((double)y1 - (double)y0);
            const double t = k * ((double)y - (double)y0);
            int32_t xi;
            if (!to_int(t + (double)x0 + 0.49999, &xi)) bad[b] = 1;
            else x = xi;
            break;
          }
        }
        row[s] = x;
      }
      pos += 2 + 4 * n;
    }
  });
  for (int b = 0; b < B; ++b)
    if (bad[b]) return GANET_ERR_NOT_FINITE;   // an x no int32 holds
  return GANET_OK;
}

int culane_text(const int32_t *lanes, int B, int threads, char *text, size_t text_bytes,
                size_t *text_start) {
  if (!lanes || !text_start) return GANET_ERR_NULL;
  if (B <= 0) return GANET_ERR_BAD_SHAPE;
  // worst case per point: two numbers of "-2147483648.00" and two separators
  text_start[0] = 0;
  for (int b = 0; b < B; ++b) {
    const int32_t *rec = lanes + (size_t)B * 4 + lanes[(size_t)b * 4 + 2];
    size_t cap = 0;
    int pos = 0;
    for (int l = 0; l < lanes[(size_t)b * 4]; ++l) {
      const int n = rec[pos + 1];
      cap += (size_t)n * 30 + 1;
      pos += 2 + 4 * n;
    }
    text_start[b + 1] = text_start[b] + cap;
  }
  if (!text || text_bytes < text_start[B]) return GANET_ERR_WORKSPACE;
  std::vector<size_t> used(B, 0);
  for_images(B, threads, (long long)(text_start[B] / 30), [&](int b) {
    const int32_t *rec = lanes + (size_t)B * 4 + lanes[(size_t)b * 4 + 2];
    char *p = text + text_start[b];
    int pos = 0;
    for (int l = 0; l < lanes[(size_t)b * 4]; ++l) {
      const int n = rec[pos + 1];
      const int32_t *pts = rec + pos + 2;
      for (int i = 0; i < n; ++i) {
        if (i) *p++ = ' ';
        p = put_int_2f(p, pts[2 * i]);
        *p++ = ' ';
        p = put_int_2f(p, pts[2 * i + 1]);
      }
      *p++ = '\n';
      pos += 2 + 4 * n;
    }
    used[b] = (size_t)(p - (text + text_start[b]));
  });
  // text_start[b] keeps the start of image b's text; its length goes to the odd slots
  for (int b = 0; b < B; ++b) text_start[B + 1 + b] = used[b];
  return GANET_OK;
}

}  // namespace

extern "C" int ganet_lanes_from_packed(const int32_t *packed, const ganet_lane_format *f,
                                       int32_t *lanes, size_t lanes_words, size_t *needed_words) {
  return guarded([&] { return lanes_from_packed(packed, f, lanes, lanes_words, needed_words); });
}

extern "C" int ganet_tusimple_from_lanes(const int32_t *lanes, int B, const int32_t *h_samples,
                                         int n_h, int reg_x, int threads, int32_t *xs,
                                         size_t xs_words, int32_t *lane_start) {
  return guarded([&] {
    return tusimple_from_lanes(lanes, B, h_samples, n_h, reg_x, threads, xs, xs_words, lane_start);
  });
}

extern "C" int ganet_culane_text(const int32_t *lanes, int B, int threads, char *text,
                                 size_t text_bytes, size_t *text_start) {
  return guarded([&] { return culane_text(lanes, B, threads, text, text_bytes, text_start); });
}
