// Keypoint decode + lane grouping for gfx950 (MI355X), hand-written HIP.
//
// Restates, for a whole batch per call,
//   GANetHeadFast.ktdet_decode_fast  mmdet/models/dense_heads/ganet_head.py:203-238
//   group_points_fast                tools/ganet/post_process.py:141-194
//
// Kernels (images up to 8192 pixels skip the first: the group kernel forms the bitmap
// in LDS itself):
//   decode_mask_kernel   grid (ceil(HW/256), B): 1x3 NMS + threshold, one pixel per
//                        lane, one 64-pixel word of the keypoint bitmap per wavefront
//                        ballot.  Reads the heat plane once, coalesced.
//   decode_group_kernel  one 256-thread workgroup per image: ordered (row-major)
//                        compaction of the bitmap by popcount + block scan, sparse
//                        gathers of offset/error at the keypoints, centre list sorted
//                        x-major by a rank sort in LDS, greedy centre clustering on one
//                        wavefront (order dependent, so serial over centres, lane
//                        parallel over earlier centres), integer means, and the
//                        first-match keypoint -> centre assignment in float64.  With
//                        DecodeArgs::packed (ganet_decode_batch_packed) it also writes
//                        the image's record of the host stream itself: one launch.
//   decode_gather_kernel / decode_assign_kernel   large by-centre images: the keypoint
//                        lists and the keypoint -> centre assignment on grids over the
//                        batch; decode_pack_kernel: the record stream as a pass of its own.
//
// Bit-exactness notes (all arithmetic below is IEEE, file built with
// -ffp-contract=off, no fast-math):
//   * thresholds are compared in float32 like torch does for Python scalars;
//   * distances use float64 sub / mul / add; `sqrt(u) <= thr` is decided as `u <= U` with
//     U = the largest double whose correctly rounded root is <= thr (sqrt_le_bound, found on
//     the host): the same decisions, no root in the loops; the nearest-centre variant, which
//     compares distances, still takes the root (correctly rounded on CDNA4);
//     the reference squares with libm pow(x, 2.0), which is 1 ulp off x*x for a
//     fraction of a percent of inexact squares - a decision can differ only when a
//     distance lies within an ulp of the threshold without being exact
//     (tests/test_oracle_golden.py::test_square_vs_libm_pow);
//   * the cluster mean is int64-sum / count in float64, truncated, like
//     np.mean(..., dtype=int).

#include "common.h"

#include <cmath>
#include <limits>
#include <mutex>

namespace {

constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / 64;
constexpr int kMaxCenters = GANET_DECODE_MAX_CENTERS;  // LDS-resident centre list
constexpr int kLdsGid = 8192;  // keypoint group ids kept in LDS (no-centre mode)
constexpr int kFusedMaskWords = 128;  // images up to 8192 pixels: bitmap formed in LDS

// heat * (max_pool2d(heat,(1,3),1,(0,1)) == heat)   ganet_head.py:205-208.
// ATen's pool starts at -inf and takes v when (v > max) || isnan(v).
// rows3: the north star's 3x3 window (max_pool2d(heat,(3,3),1,(1,1)), walked row-major
// like ATen); the reference always runs 1x3.  `img` is the image's first row.
// flags bit3: the plane holds LOGITS and every value read goes through
//   hm = torch.clamp(hm.sigmoid(), min=1e-4, max=1 - 1e-4)      ganet_head.py:291
// first: 1 / (1 + exp(-v)) in float32 as ATen's GPU sigmoid forms it, then clamp's
// min/max order with NaN kept.
__device__ __forceinline__ float heat_prob(float v) {
  const float s = 1.0f / (1.0f + expf(-v));
  const float lo = 1e-4f, hi = (float)(1 - 1e-4);
  return s != s ? s : fminf(fmaxf(s, lo), hi);
}

template <bool LOGITS>
__device__ __forceinline__ float heat_nms(const float *__restrict__ img, int H, int W,
                                          int y, int x, bool rows3) {
  auto rd = [](float v) { return LOGITS ? heat_prob(v) : v; };
  const float h = rd(img[y * W + x]);
  float m = -INFINITY;
  const int y0 = rows3 && y > 0 ? y - 1 : y;
  const int y1 = rows3 && y < H - 1 ? y + 1 : y;
  for (int yy = y0; yy <= y1; ++yy) {
    const float *row = img + yy * W;
    if (x > 0) {
      const float v = rd(row[x - 1]);
      if (v > m || v != v) m = v;
    }
    {
      const float v = rd(row[x]);
      if (v > m || v != v) m = v;
    }
    if (x < W - 1) {
      const float v = rd(row[x + 1]);
      if (v > m || v != v) m = v;
    }
  }
  const float keep = (m == h) ? 1.0f : 0.0f;
  return h * keep;
}

__global__ __launch_bounds__(kBlock) void decode_mask_kernel(
    const float *__restrict__ heat, int HW, int W, float thr,
    unsigned long long *__restrict__ kmask, int nwords, int rows3, int logits,
    int32_t *__restrict__ counts) {
  const int b = blockIdx.y;
  const int idx = blockIdx.x * kBlock + threadIdx.x;
  // large images: decode_gather_kernel counts the centres in counts[b][1] by atomics
  if (counts && blockIdx.x == 0 && threadIdx.x < GANET_DECODE_COUNT_STRIDE)
    counts[(size_t)b * GANET_DECODE_COUNT_STRIDE + threadIdx.x] = 0;
  bool is_kpt = false;
  if (idx < HW) {
    const int y = idx / W;
    const int x = idx - y * W;
    const float *img = heat + (size_t)b * HW;
    is_kpt = (logits ? heat_nms<true>(img, HW / W, W, y, x, rows3 != 0)
                     : heat_nms<false>(img, HW / W, W, y, x, rows3 != 0)) > thr;
  }
  const unsigned long long m = __ballot(is_kpt);
  const int word = idx >> 6;
  if ((threadIdx.x & 63) == 0 && word < nwords)
    kmask[(size_t)b * nwords + word] = m;
}

// Wavefront scan / reduction on the DPP path (row shifts inside the 16-lane rows, then the two
// row broadcasts of GFX9): six VALU instructions instead of six trips through the LDS
// crossbar (ds_bpermute), which is what a __shfl costs.
template <typename OP>
__device__ __forceinline__ int wave_scan_dpp(int v, int identity, OP op) {
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x111, 0xf, 0xf, false));  // row_shr:1
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x112, 0xf, 0xf, false));  // row_shr:2
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x114, 0xf, 0xf, false));  // row_shr:4
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x118, 0xf, 0xf, false));  // row_shr:8
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x142, 0xa, 0xf, false));  // row_bcast:15 -> rows 1, 3
  v = op(v, __builtin_amdgcn_update_dpp(identity, v, 0x143, 0xc, 0xf, false));  // row_bcast:31 -> rows 2, 3
  return v;   // inclusive over lanes 0..lane
}

// Exclusive scan of one int per thread over the 256-thread block.
__device__ __forceinline__ int block_excl_scan(int v, int *wave_sums,
                                               int *total) {
  const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
  const int incl = wave_scan_dpp(v, 0, [](int a, int b) { return a + b; });
  if (lane == 63) wave_sums[wid] = incl;
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < kWavesPerBlock; ++w) {
    const int s = wave_sums[w];
    if (w < wid) off += s;
    tot += s;
  }
  __syncthreads();
  *total = tot;
  return off + incl - v;
}

__device__ __forceinline__ int wave_min(int v) {
  v = wave_scan_dpp(v, INT_MAX, [](int a, int b) { return min(a, b); });
  return __builtin_amdgcn_readlane(v, 63);
}

// `sqrt(u) <= T` decided without the square root.  The correctly rounded square root is
// monotone, so the u >= 0 that pass are exactly the doubles up to one bound; the bound is found
// once per call on the host (T*T, then a step or two along the doubles, with the same IEEE
// sqrt), so every decision is the one `sqrt(u) <= T` makes - NaN and infinite u included.
// -1: nothing passes (T negative or NaN).
double sqrt_le_bound(double T) {
  if (!(T >= 0.0)) return -1.0;
  if (std::isinf(T)) return T;
  double u = T * T;
  if (std::isinf(u)) u = std::numeric_limits<double>::max();
  while (u > 0.0 && std::sqrt(u) > T) u = std::nextafter(u, 0.0);
  for (;;) {
    const double n = std::nextafter(u, std::numeric_limits<double>::infinity());
    if (!(std::sqrt(n) <= T)) break;
    u = n;
  }
  return u;
}

#ifdef GANET_DECODE_STAMPS   // profiling build only (tools/decode_stamps.py): phase times of image 0
__device__ unsigned long long g_decode_stamps[16];
#define DSTAMP(i)                                                            \
  do {                                                                       \
    if (blockIdx.x == 0 && threadIdx.x == 0) g_decode_stamps[i] = wall_clock64(); \
  } while (0)
#else
#define DSTAMP(i) \
  do {            \
  } while (0)
#endif

struct DecodeArgs {
  const float *heat, *offset, *error;
  const unsigned long long *kmask;   // nullptr: the group kernel forms the bitmap itself
  float kpt_thr;
  int H, W, nwords;
  float root_thr;
  double cluster_thr, by_center_thr;
  double cluster_sq, by_center_sq;   // sqrt_le_bound of the two: `d <= thr` as `d*d <= .._sq`
  int max_kpts, max_centers, max_groups, flags;
  ganet_decode_out out;
  // ganet_decode_batch_packed, one-workgroup form: the record stream of ganet_decode_pack is
  // written by the group kernel itself (nullptr: not)
  int32_t *packed;
  long long packed_words;
  unsigned long long *pack_ctr;      // [0] words handed out so far, [1] images finished
  int B;
};

// Large images (by-centre grouping): the keypoint lists of the whole batch on a grid of
// workgroups instead of one workgroup per image.  Workgroup (chunk, image) owns 256 words of
// the image's bitmap: it sums the popcounts of the words before its chunk (the keypoints'
// row-major rank is an ORDERED count, no atomics), scans its own, and every thread gathers
// the keypoints of its word.  Centres (offset_y < root_thr, ganet_head.py:213-216) go
// unordered into centers_xy[slot] as packed (x << 16 | y) keys counted in counts[b][1]; the
// group kernel rank-sorts them x-major, so their arrival order does not matter.
__global__ __launch_bounds__(kBlock) void decode_gather_kernel(DecodeArgs a) {
  __shared__ int wave_sums[kWavesPerBlock];
  __shared__ int s_base;
  const int b = blockIdx.y, chunk = blockIdx.x;
  const int tid = threadIdx.x;
  const int W = a.W;
  const size_t HW = (size_t)a.H * W;
  const float *off_x = a.offset + (size_t)b * 2 * HW;
  const float *off_y = off_x + HW;
  const float *err_x = a.error + (size_t)b * 2 * HW;
  const float *err_y = err_x + HW;
  const unsigned long long *kmask = a.kmask + (size_t)b * a.nwords;
  int32_t *o_counts = a.out.counts + (size_t)b * GANET_DECODE_COUNT_STRIDE;
  int32_t *o_cxy = a.out.centers_xy + (size_t)b * a.max_centers * 2;
  int32_t *o_kxy = a.out.kpt_xy + (size_t)b * a.max_kpts * 2;
  float *o_align = a.out.kpt_align + (size_t)b * a.max_kpts * 2;
  float *o_root = a.out.kpt_root + (size_t)b * a.max_kpts * 2;
  int32_t *o_kgid = a.out.kpt_gid + (size_t)b * a.max_kpts;

  // keypoints in the words before this chunk
  int before = 0;
  for (int w = tid; w < chunk * kBlock; w += kBlock) before += __popcll(kmask[w]);
  int total;
  const int ex = block_excl_scan(before, wave_sums, &total);
  (void)ex;
  if (tid == 0) s_base = total;
  __syncthreads();
  const int base = s_base;
  __syncthreads();

  const int w = chunk * kBlock + tid;
  unsigned long long bits = (w < a.nwords) ? kmask[w] : 0ull;
  int pos = base + block_excl_scan(__popcll(bits), wave_sums, &total);
  if (chunk == (int)gridDim.x - 1 && tid == 0) o_counts[0] = base + total;   // n_kpts
  while (bits) {
    const int bit = __ffsll((long long)bits) - 1;
    bits &= bits - 1;
    const int idx = w * 64 + bit;
    const int y = idx / W, x = idx - y * W;
    const float oy = off_y[idx], ox = off_x[idx], exx = err_x[idx], eyy = err_y[idx];
    if (pos < a.max_kpts) {
      const float fx = (float)x, fy = (float)y;
      o_kxy[2 * pos] = x;
      o_kxy[2 * pos + 1] = y;
      o_root[2 * pos] = fx + ox;      // ganet_head.py:229
      o_root[2 * pos + 1] = fy + oy;
      o_align[2 * pos] = fx + exx;    // ganet_head.py:230
      o_align[2 * pos + 1] = fy + eyy;
      o_kgid[pos] = -1;
    }
    if (oy < a.root_thr) {
      const int slot = atomicAdd(&o_counts[1], 1);
      if (slot < a.max_centers) o_cxy[2 * slot] = (int)(((unsigned int)x << 16) | (unsigned int)y);
    }
    ++pos;
  }
}

// Large images, after the group kernel: every keypoint to the first (flags bit2: nearest)
// group centre within thr - step 5 of decode_group_kernel on a grid over the batch.
__global__ __launch_bounds__(kBlock) void decode_assign_kernel(DecodeArgs a) {
  __shared__ int s_gx[kBlock], s_gy[kBlock];
  const int b = blockIdx.y;
  const int tid = threadIdx.x;
  const int32_t *o_counts = a.out.counts + (size_t)b * GANET_DECODE_COUNT_STRIDE;
  const float *o_root = a.out.kpt_root + (size_t)b * a.max_kpts * 2;
  int32_t *o_kgid = a.out.kpt_gid + (size_t)b * a.max_kpts;
  const int32_t *o_gc = a.out.group_center + (size_t)b * a.max_groups * 2;
  int32_t *o_gn = a.out.group_count + (size_t)b * a.max_groups;
  const int nk_st = min(o_counts[0], a.max_kpts);
  const int ng = min(o_counts[2], a.max_groups);   // more groups: flagged as overflow
  if ((int)blockIdx.x * kBlock >= nk_st) return;    // uniform
  const double T = a.by_center_thr;
  const double U = a.by_center_sq;
  const bool nearest = (a.flags & 4) != 0;
  const int i = blockIdx.x * kBlock + tid;
  const bool live = i < nk_st;
  const double rx = live ? (double)o_root[2 * i] : 0.0, ry = live ? (double)o_root[2 * i + 1] : 0.0;
  int found = -1;
  double best = 0;
  for (int g0 = 0; g0 < ng; g0 += kBlock) {
    __syncthreads();
    if (g0 + tid < ng) {
      s_gx[tid] = o_gc[2 * (g0 + tid)];
      s_gy[tid] = o_gc[2 * (g0 + tid) + 1];
    }
    __syncthreads();
    const int n = min(kBlock, ng - g0);
    for (int g = 0; g < n; ++g) {
      if (found >= 0 && !nearest) break;
      const double dx = rx - (double)s_gx[g];
      const double dy = ry - (double)s_gy[g];
      const double ss = dx * dx + dy * dy;
      if (!nearest) {
        if (ss <= U) found = g0 + g;      // = sqrt(ss) <= T
        continue;
      }
      const double d = sqrt(ss);
      if (d <= T && (found < 0 || d < best)) {
        found = g0 + g;
        best = d;
      }
    }
  }
  if (live) o_kgid[i] = found;
  // group sizes: a histogram in LDS per workgroup, then one global add per group (the
  // keypoints of a lane are neighbours, so most of a workgroup adds to one or two groups)
  if (ng <= kBlock) {
    __syncthreads();
    s_gx[tid] = 0;
    __syncthreads();
    if (live && found >= 0) atomicAdd(&s_gx[found], 1);
    __syncthreads();
    if (tid < ng && s_gx[tid]) atomicAdd(&o_gn[tid], s_gx[tid]);
  } else if (live && found >= 0) {
    atomicAdd(&o_gn[found], 1);
  }
}

// The record stream written from inside decode_group_kernel<0> (DecodeArgs::packed).  Same
// header and record layout as decode_pack_kernel below, but a record's place in the stream is
// handed out by an atomic counter when the image's sizes are known, so records follow one
// another in the order the images got there - readers go by the header's offsets - and the
// words are stored where they are produced: no second launch, no second pass over the lists.
// The last workgroup to finish writes the total and leaves both counters at zero for the next
// call on this workspace.
struct RecordSlot {
  int32_t *r;        // the image's record, nullptr: no stream asked for or the record does not fit
  int off, words;    // header words 4 and 5
  bool fits;
};

// MODE 0: build the keypoint / centre lists from the bitmap (steps 1-2), group, assign.
// MODE 1: the lists already sit in the output arrays (ganet_group_batch).
// MODE 2: large images - decode_gather_kernel wrote the keypoints and the unsorted centre
//         keys; sort and cluster the centres here, decode_assign_kernel assigns afterwards.
template <int MODE>
__global__ __launch_bounds__(kBlock) void decode_group_kernel(DecodeArgs a) {
  constexpr bool DECODE = MODE == 0;
  // One LDS carve-up for both grouping modes (48 KiB).
  __shared__ int smem[6 * kMaxCenters];
  __shared__ int wave_sums[kWavesPerBlock];
  __shared__ int s_nc, s_ng;
  __shared__ unsigned long long s_rec;
  __shared__ unsigned long long s_mask[DECODE ? kFusedMaskWords : 1];
  __shared__ __attribute__((aligned(16))) float s_heat[DECODE ? kFusedMaskWords * 64 : 1];

  unsigned int *ckey_raw = reinterpret_cast<unsigned int *>(smem);  // unsorted keys
  unsigned int *ckey = ckey_raw + kMaxCenters;                      // x-major keys
  volatile int *cgid = smem + 2 * kMaxCenters;
  int *gsumx = smem + 3 * kMaxCenters;
  int *gsumy = smem + 4 * kMaxCenters;
  int *gcnt = smem + 5 * kMaxCenters;

  const int b = blockIdx.x;
  const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
  const int H = a.H, W = a.W;
  const size_t HW = (size_t)H * W;
  const float *off_x = a.offset + (size_t)b * 2 * HW;
  const float *off_y = off_x + HW;
  const float *err_x = a.error + (size_t)b * 2 * HW;
  const float *err_y = err_x + HW;
  const unsigned long long *kmask = a.kmask ? a.kmask + (size_t)b * a.nwords : nullptr;

  int32_t *o_counts = a.out.counts + (size_t)b * GANET_DECODE_COUNT_STRIDE;
  int32_t *o_cxy = a.out.centers_xy + (size_t)b * a.max_centers * 2;
  int32_t *o_cgid = a.out.center_gid + (size_t)b * a.max_centers;
  int32_t *o_kxy = a.out.kpt_xy + (size_t)b * a.max_kpts * 2;
  float *o_align = a.out.kpt_align + (size_t)b * a.max_kpts * 2;
  float *o_root = a.out.kpt_root + (size_t)b * a.max_kpts * 2;
  int32_t *o_kgid = a.out.kpt_gid + (size_t)b * a.max_kpts;
  int32_t *o_gc = a.out.group_center + (size_t)b * a.max_groups * 2;
  int32_t *o_gn = a.out.group_count + (size_t)b * a.max_groups;

  if (tid == 0) {
    s_nc = 0;
    s_ng = 0;
  }
  __syncthreads();
  DSTAMP(0);

  int nk = 0, nc_found = 0, nk_st = 0, nc = 0, overflow = 0;
  if constexpr (DECODE) {
  // ---- 0. small images: NMS + threshold here instead of in decode_mask_kernel
  //         (one launch less).  The heat plane is staged in LDS with all loads in
  //         flight at once; the bitmap never leaves LDS. --------------------------
  const bool fused_mask = kmask == nullptr;
  if (fused_mask) {
    const float *hb = a.heat + (size_t)b * HW;
    const bool logits = (a.flags & 8) != 0;   // probabilities formed once, on the way into LDS
    if ((HW & 3) == 0) {
      // 16-byte loads, 8 per thread in flight (8192 pixels = 2048 float4)
      const float4 *hb4 = reinterpret_cast<const float4 *>(hb);
      float4 *sh4 = reinterpret_cast<float4 *>(s_heat);
      const int n4 = (int)(HW >> 2);
      float4 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int q = tid + k * kBlock;
        v[k] = hb4[q < n4 ? q : 0];
      }
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int q = tid + k * kBlock;
        if (logits)
          v[k] = make_float4(heat_prob(v[k].x), heat_prob(v[k].y), heat_prob(v[k].z),
                             heat_prob(v[k].w));
        if (q < n4) sh4[q] = v[k];
      }
    } else {
      for (int p = tid; p < (int)HW; p += kBlock)
        s_heat[p] = logits ? heat_prob(hb[p]) : hb[p];
    }
    if (!(a.flags & 2)) {
      // the reference's 1x3 window: 16 consecutive pixels per thread and pass (18 values read
      // at once, one division), their 16 bits OR-ed into the bitmap - the decisions of
      // heat_nms, without its branches
      unsigned int *m32 = reinterpret_cast<unsigned int *>(s_mask);
      for (int w = tid; w < 2 * a.nwords; w += kBlock) m32[w] = 0u;
      __syncthreads();            // (also: the heat plane is in LDS)
      DSTAMP(1);
      const int nchunk = ((int)HW + 15) >> 4;
      for (int c = tid; c < nchunk; c += kBlock) {
        const int p0 = c << 4;
        float v[18];
        const float4 *q = reinterpret_cast<const float4 *>(s_heat + p0);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const float4 t = q[k];
          v[1 + 4 * k] = t.x;
          v[2 + 4 * k] = t.y;
          v[3 + 4 * k] = t.z;
          v[4 + 4 * k] = t.w;
        }
        v[0] = s_heat[p0 > 0 ? p0 - 1 : 0];
        v[17] = s_heat[min(p0 + 16, kFusedMaskWords * 64 - 1)];
        int x = p0 % W;
        unsigned int bits = 0u;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
          // heat_nms's running maximum m (take v when v > m or v is NaN) ends equal to h
          // exactly when h >= both neighbours inside the row and none of the three is NaN
          // (a NaN makes m NaN or a comparison false; -0 == +0 either way), so `m == h` is
          // these two comparisons, with a neighbour outside the row standing at -inf
          const float h = v[j + 1];
          const float l = (x == 0) ? -INFINITY : v[j];
          const float r = (x == W - 1) ? -INFINITY : v[j + 2];
          const float keep = (h >= l && h >= r) ? 1.0f : 0.0f;
          if (p0 + j < (int)HW && h * keep > a.kpt_thr) bits |= 1u << j;
          x = (x + 1 == W) ? 0 : x + 1;
        }
        if (bits) atomicOr(&m32[c >> 1], bits << ((c & 1) << 4));
      }
    } else {
    __syncthreads();
    DSTAMP(1);
    for (int p0 = 0; p0 < (int)HW; p0 += kBlock) {
      const int idx = p0 + tid;
      bool is_kpt = false;
      if (idx < (int)HW) {
        const int y = idx / W;
        const int x = idx - y * W;
        is_kpt = heat_nms<false>(s_heat, a.H, W, y, x, true) > a.kpt_thr;
      }
      const unsigned long long m = __ballot(is_kpt);
      if (lane == 0 && (idx >> 6) < a.nwords) s_mask[idx >> 6] = m;
    }
    }
    __syncthreads();
    DSTAMP(2);
  }
  if (fused_mask) {
    // ---- 1a. small images: ordered compaction of the pixel indices into LDS (the
    //          heat copy is dead by now), then one keypoint per thread with its four
    //          gathers in flight together - two memory round trips per image instead
    //          of two per set bit. -------------------------------------------------
    int *s_kidx = reinterpret_cast<int *>(s_heat);
    for (int w0 = 0; w0 < a.nwords; w0 += kBlock) {
      const int w = w0 + tid;
      unsigned long long bits = (w < a.nwords) ? s_mask[w] : 0ull;
      int total;
      int pos = nk + block_excl_scan(__popcll(bits), wave_sums, &total);
      nk += total;
      while (bits) {
        const int bit = __ffsll((long long)bits) - 1;
        bits &= bits - 1;
        s_kidx[pos++] = w * 64 + bit;   // pos < HW <= capacity of s_heat
      }
    }
    __syncthreads();
    DSTAMP(3);
    for (int pos = tid; pos < nk; pos += kBlock) {
      const int idx = s_kidx[pos];
      const int y = idx / W, x = idx - y * W;
      const float oy = off_y[idx], ox = off_x[idx], ex = err_x[idx], ey = err_y[idx];
      if (pos < a.max_kpts) {
        const float fx = (float)x, fy = (float)y;
        o_kxy[2 * pos] = x;
        o_kxy[2 * pos + 1] = y;
        o_root[2 * pos] = fx + ox;      // ganet_head.py:229
        o_root[2 * pos + 1] = fy + oy;
        o_align[2 * pos] = fx + ex;     // ganet_head.py:230
        o_align[2 * pos + 1] = fy + ey;
        o_kgid[pos] = -1;
      }
      if (oy < a.root_thr) {            // centre (ganet_head.py:213-216)
        const int slot = atomicAdd(&s_nc, 1);
        if (slot < kMaxCenters)
          ckey_raw[slot] = ((unsigned int)x << 16) | (unsigned int)y;
      }
    }
  } else
  // ---- 1. ordered compaction of keypoints (row-major) ---------------------
  for (int w0 = 0; w0 < a.nwords; w0 += kBlock) {
    const int w = w0 + tid;
    unsigned long long bits = (w < a.nwords) ? kmask[w] : 0ull;
    int total;
    int pos = nk + block_excl_scan(__popcll(bits), wave_sums, &total);
    nk += total;
    while (bits) {
      const int bit = __ffsll((long long)bits) - 1;
      bits &= bits - 1;
      const int idx = w * 64 + bit;
      const int y = idx / W, x = idx - y * W;
      const float oy = off_y[idx];
      if (pos < a.max_kpts) {
        const float fx = (float)x, fy = (float)y;
        o_kxy[2 * pos] = x;
        o_kxy[2 * pos + 1] = y;
        o_root[2 * pos] = fx + off_x[idx];      // ganet_head.py:229
        o_root[2 * pos + 1] = fy + oy;
        o_align[2 * pos] = fx + err_x[idx];     // ganet_head.py:230
        o_align[2 * pos + 1] = fy + err_y[idx];
        o_kgid[pos] = -1;
      }
      // centre: offset_y < root_thr on a keypoint  (ganet_head.py:213-216)
      if (oy < a.root_thr) {
        const int slot = atomicAdd(&s_nc, 1);
        if (slot < kMaxCenters)
          ckey_raw[slot] = ((unsigned int)x << 16) | (unsigned int)y;
      }
      ++pos;
    }
  }
  __syncthreads();
  DSTAMP(4);
  nc_found = s_nc;
  nk_st = min(nk, a.max_kpts);
  nc = min(nc_found, min(a.max_centers, kMaxCenters));
  overflow = (nk > a.max_kpts ? 1 : 0) |
             (nc_found > min(a.max_centers, kMaxCenters) ? 2 : 0);

  // ---- 2. centres in x-major order: rank sort on (x, y) -------------------
  // (the reference transposes the mask to [W,H] before np.where,
  //  ganet_head.py:217-219).  Keys are distinct pixels, so ranks are unique.
  const int nc_sort = min(nc_found, kMaxCenters);
  for (int i = tid; i < nc_sort; i += kBlock) {
    const unsigned int ki = ckey_raw[i];
    int rank = 0;
    for (int j = 0; j < nc_sort; ++j) rank += (ckey_raw[j] < ki);
    ckey[rank] = ki;
  }
  __syncthreads();
  for (int i = tid; i < nc; i += kBlock) {
    o_cxy[2 * i] = (int)(ckey[i] >> 16);
    o_cxy[2 * i + 1] = (int)(ckey[i] & 0xffffu);
  }

  } else if constexpr (MODE == 2) {
    nk = o_counts[0];
    nc_found = o_counts[1];
    nk_st = min(nk, a.max_kpts);
    nc = min(nc_found, min(a.max_centers, kMaxCenters));
    overflow = (nk > a.max_kpts ? 1 : 0) |
               (nc_found > min(a.max_centers, kMaxCenters) ? 2 : 0);
    // x-major order by rank (ganet_head.py:217-219); keys are distinct pixels
    for (int i = tid; i < nc; i += kBlock) ckey_raw[i] = (unsigned int)o_cxy[2 * i];
    __syncthreads();
    for (int i = tid; i < nc; i += kBlock) {
      const unsigned int ki = ckey_raw[i];
      int rank = 0;
      for (int j = 0; j < nc; ++j) rank += (ckey_raw[j] < ki);
      ckey[rank] = ki;
    }
    __syncthreads();
    for (int i = tid; i < nc; i += kBlock) {
      o_cxy[2 * i] = (int)(ckey[i] >> 16);
      o_cxy[2 * i + 1] = (int)(ckey[i] & 0xffffu);
    }
  } else {
    nk = o_counts[0];
    nc_found = o_counts[1];
    overflow = o_counts[3] & 3;
    nk_st = min(nk, a.max_kpts);
    nc = min(nc_found, min(a.max_centers, kMaxCenters));
    for (int i = tid; i < nc; i += kBlock)
      ckey[i] = ((unsigned int)o_cxy[2 * i] << 16) | (unsigned int)o_cxy[2 * i + 1];
    __syncthreads();
  }
  DSTAMP(5);    // every mode: the centre list stands sorted in LDS
  int ng = 0;
  RecordSlot rec = {nullptr, 0, 0, true};
  // block-uniform; called once, when the number of groups is known
  auto take_record = [&](int ng_now) {
    if constexpr (DECODE) {
      if (!a.packed) return;
      const int ngs = min(ng_now, a.max_groups);
      rec.words = 7 * nk_st + 3 * nc + (a.by_center_thr >= 0 ? 2 * ngs : 0) + ngs;
      unsigned long long off = 0;          // a batch of one: the record starts the stream
      if (a.B > 1) {
        if (tid == 0) s_rec = atomicAdd(a.pack_ctr, (unsigned long long)rec.words);
        __syncthreads();
        off = s_rec;
      }
      rec.fits = (unsigned long long)a.B * 8 + off + (unsigned long long)rec.words <=
                 (unsigned long long)a.packed_words;
      rec.off = (int)(off < 0x7fffffffull ? off : 0x7fffffffull);
      rec.r = rec.fits ? a.packed + (size_t)a.B * 8 + off : nullptr;
    }
  };
  // keypoint i of the record: the lists were stored by this workgroup before a barrier
  auto put_kpt = [&](int i, int gid) {
    const int2 xy = *reinterpret_cast<const int2 *>(o_kxy + 2 * i);
    const int2 al = *reinterpret_cast<const int2 *>(o_align + 2 * i);
    const int2 rt = *reinterpret_cast<const int2 *>(o_root + 2 * i);
    int32_t *r = rec.r;
    r[2 * i] = xy.x;
    r[2 * i + 1] = xy.y;
    r[2 * nk_st + 2 * i] = al.x;
    r[2 * nk_st + 2 * i + 1] = al.y;
    r[4 * nk_st + 2 * i] = rt.x;
    r[4 * nk_st + 2 * i + 1] = rt.y;
    r[6 * nk_st + i] = gid;
  };
  if (a.flags & 1) {
    // decode only
    take_record(0);
    if (rec.r) {
      for (int i = tid; i < nk_st; i += kBlock) put_kpt(i, -1);
      for (int i = tid; i < nc; i += kBlock) {
        rec.r[7 * nk_st + 2 * i] = o_cxy[2 * i];
        rec.r[7 * nk_st + 2 * i + 1] = o_cxy[2 * i + 1];
        rec.r[7 * nk_st + 2 * nc + i] = o_cgid[i];
      }
    }
  } else if (a.by_center_thr >= 0) {
    // ---- 3. greedy clustering of the centres (post_process.py:160-165) ----
    // search_groups returns the first group holding ANY member within thr,
    // i.e. the smallest group index among earlier centres within thr.
    const double T = a.by_center_thr;
    const double U = a.by_center_sq;
    if (wid == 0 && nc <= 64) {
      // up to 64 centres (every lane image the configs produce): lane j keeps centre j and,
      // as lane g, the members of group g as a 64-bit set.  Centre i joins the lowest group
      // that holds an earlier centre within thr: one ballot for "who is near", one for "which
      // groups hold one of those", find-first-set - no reduction, no LDS in the loop.
      const unsigned int kj = lane < nc ? ckey[lane] : 0u;
      const int xj = (int)(kj >> 16), yj = (int)(kj & 0xffffu);
      unsigned long long members = 0ull;
      int mine = 0, ngl = 0;
      for (int i = 0; i < nc; ++i) {
        const int xi = __builtin_amdgcn_readlane(xj, i), yi = __builtin_amdgcn_readlane(yj, i);
        const double dx = (double)(xi - xj), dy = (double)(yi - yj);
        const unsigned long long near = __ballot(lane < i && dx * dx + dy * dy <= U);
        const unsigned long long hit = __ballot((members & near) != 0ull);
        const int g = hit ? __ffsll((long long)hit) - 1 : ngl++;
        if (lane == g) members |= 1ull << i;
        if (lane == i) mine = g;
      }
      if (lane < nc) cgid[lane] = mine;
      if (lane == 0) s_ng = ngl;
    } else if (wid == 0) {
      int ngl = 0;
      for (int i = 0; i < nc; ++i) {
        const unsigned int ki = ckey[i];
        const int xi = (int)(ki >> 16), yi = (int)(ki & 0xffffu);
        int best = INT_MAX;
        for (int j = lane; j < i; j += 64) {
          const unsigned int kj = ckey[j];
          const double dx = (double)(xi - (int)(kj >> 16));
          const double dy = (double)(yi - (int)(kj & 0xffffu));
          if (dx * dx + dy * dy <= U) best = min(best, (int)cgid[j]);   // = sqrt(.) <= T
        }
        best = wave_min(best);
        const int g = (best == INT_MAX) ? ngl++ : best;
        if (lane == 0) cgid[i] = g;
      }
      if (lane == 0) s_ng = ngl;
    }
    __syncthreads();
    DSTAMP(6);
    ng = s_ng;
    take_record(ng);
    DSTAMP(7);
    int32_t *const rc_ = rec.r ? rec.r + 7 * nk_st : nullptr;          // centres [2nc], ids [nc]
    int32_t *const rg_ = rec.r ? rec.r + 7 * nk_st + 3 * nc : nullptr;  // group centres, counts
    // ---- 4. truncated integer mean per group (post_process.py:38-41) ------
    for (int g = tid; g < ng; g += kBlock) {
      gsumx[g] = 0;
      gsumy[g] = 0;
      gcnt[g] = 0;
    }
    __syncthreads();
    for (int i = tid; i < nc; i += kBlock) {
      const int g = cgid[i];
      atomicAdd(&gsumx[g], (int)(ckey[i] >> 16));
      atomicAdd(&gsumy[g], (int)(ckey[i] & 0xffffu));
      atomicAdd(&gcnt[g], 1);
      o_cgid[i] = g;
      if (rc_) {
        rc_[2 * i] = (int)(ckey[i] >> 16);
        rc_[2 * i + 1] = (int)(ckey[i] & 0xffffu);
        rc_[2 * nc + i] = g;
      }
    }
    __syncthreads();
    for (int g = tid; g < ng; g += kBlock) {
      const double c = (double)gcnt[g];
      const int mx = (int)(long long)((double)gsumx[g] / c);
      const int my = (int)(long long)((double)gsumy[g] / c);
      gsumx[g] = mx;
      gsumy[g] = my;
      gcnt[g] = 0;
      if (g < a.max_groups) {
        o_gc[2 * g] = mx;
        o_gc[2 * g + 1] = my;
        if (rg_) {
          rg_[2 * g] = mx;
          rg_[2 * g + 1] = my;
        }
      }
    }
    __syncthreads();
    DSTAMP(8);
    // ---- 5. keypoint -> first centre within thr (post_process.py:61-66,173-179);
    //         flags bit2: the north star's variant, the NEAREST centre within thr
    //         (ties to the lower group index)
    const bool nearest = (a.flags & 4) != 0;
    if constexpr (MODE == 2) {
      for (int g = tid; g < min(ng, a.max_groups); g += kBlock) o_gn[g] = 0;
    } else {
    for (int i = tid; i < nk_st; i += kBlock) {
      const double rx = (double)o_root[2 * i], ry = (double)o_root[2 * i + 1];
      int found = -1;
      double best = 0;
      for (int g = 0; g < ng; ++g) {
        const double dx = rx - (double)gsumx[g];
        const double dy = ry - (double)gsumy[g];
        const double ss = dx * dx + dy * dy;
        if (!nearest) {
          if (ss <= U) {                    // = sqrt(ss) <= T
            found = g;
            break;
          }
          continue;
        }
        const double d = sqrt(ss);
        if (d <= T && (found < 0 || d < best)) {
          found = g;
          best = d;
        }
      }
      o_kgid[i] = found;
      if (found >= 0) atomicAdd(&gcnt[found], 1);
      if (rec.r) put_kpt(i, found);
    }
    __syncthreads();
    DSTAMP(9);
    for (int g = tid; g < min(ng, a.max_groups); g += kBlock) {
      o_gn[g] = gcnt[g];
      if (rg_) rg_[2 * min(ng, a.max_groups) + g] = gcnt[g];
    }
    }
  } else {
    // ---- no centre clustering: update_coords_fast (post_process.py:142-153)
    // Greedy over keypoints in order; group ids live in LDS while they fit,
    // else in the (workgroup-coherent) output array.
    const double U = a.cluster_sq;
    __syncthreads();  // o_root written above by the whole block
    const bool in_lds = nk_st <= kLdsGid;
    volatile int *kg = in_lds ? (volatile int *)smem : (volatile int *)o_kgid;
    if (wid == 0) {
      int ngl = 0;
      for (int i = 0; i < nk_st; ++i) {
        const double rx = (double)o_root[2 * i], ry = (double)o_root[2 * i + 1];
        int best = INT_MAX;
        for (int j = lane; j < i; j += 64) {
          const double dx = rx - (double)o_root[2 * j];
          const double dy = ry - (double)o_root[2 * j + 1];
          if (dx * dx + dy * dy <= U) best = min(best, (int)kg[j]);      // = sqrt(.) <= cluster_thr
        }
        best = wave_min(best);
        const int g = (best == INT_MAX) ? ngl++ : best;
        if (lane == 0) kg[i] = g;
        if (!in_lds) __threadfence_block();
      }
      if (lane == 0) s_ng = ngl;
    }
    __syncthreads();
    ng = s_ng;
    for (int g = tid; g < min(ng, a.max_groups); g += kBlock) o_gn[g] = 0;
    __syncthreads();
    for (int i = tid; i < nk_st; i += kBlock) {
      const int g = kg[i];
      if (in_lds) o_kgid[i] = g;
      if (g < a.max_groups) atomicAdd(&o_gn[g], 1);
    }
    take_record(ng);     // (its barrier also orders the atomics above before the reads below)
    if (rec.r) {
      for (int i = tid; i < nk_st; i += kBlock) put_kpt(i, kg[i]);
      // the centre list sits in the output arrays (kg may have taken its place in LDS);
      // centre ids are not formed in this mode: the array's words go out as they are,
      // like decode_pack_kernel's copy
      for (int i = tid; i < nc; i += kBlock) {
        rec.r[7 * nk_st + 2 * i] = o_cxy[2 * i];
        rec.r[7 * nk_st + 2 * i + 1] = o_cxy[2 * i + 1];
        rec.r[7 * nk_st + 2 * nc + i] = o_cgid[i];
      }
      for (int g = tid; g < min(ng, a.max_groups); g += kBlock)
        rec.r[7 * nk_st + 3 * nc + g] = (int)atomicAdd(&o_gn[g], 0);   // read where it was added
    }
  }
  if (ng > a.max_groups) overflow |= 4;
  if (tid == 0) {
    o_counts[0] = nk;
    o_counts[1] = nc_found;
    o_counts[2] = ng;
    o_counts[3] = overflow;
    o_counts[4] = 0;
    o_counts[5] = 0;
    o_counts[6] = 0;
    o_counts[7] = 0;
  }
  if constexpr (DECODE) {
    if (a.packed && tid == 0) {
      int32_t *hdr = a.packed + (size_t)b * 8;
      hdr[0] = nk;
      hdr[1] = nc_found;
      hdr[2] = ng;
      hdr[3] = overflow | (rec.fits ? 0 : 8);
      hdr[4] = rec.off;
      hdr[5] = rec.words;
      if (b != 0) hdr[6] = 0;
      hdr[7] = 0;
      // the last image to get here knows the total (every image took its words before)
      if (a.B == 1) {
        a.packed[6] = rec.words;
      } else if (atomicAdd(a.pack_ctr + 1, 1ull) == (unsigned long long)(a.B - 1)) {
        const unsigned long long total = atomicExch(a.pack_ctr, 0ull);
        atomicExch(a.pack_ctr + 1, 0ull);
        a.packed[6] = (int)(total < 0x7fffffffull ? total : 0x7fffffffull);
      }
    }
  }
  DSTAMP(10);
}

// Compacts the capacity-strided per-image arrays into one dense record stream so
// that the host copy moves only what was found:
//   packed = [header B x 8][record 0][record 1]...
//   header[b] = {n_kpts, n_centers, n_groups, overflow, record offset (words,
//                from the end of the header), record words, total words (b == 0
//                only), 0}
//   record = kpt_xy[2nk] align[2nk] root[2nk] kpt_gid[nk] centers[2nc]
//            center_gid[nc] group_center[2ngc] group_count[ng]
// with nk/nc/ng clamped to the capacities and ngc = ng in by-centre mode, else 0.
__device__ __forceinline__ int record_words(const int32_t *c, int max_k, int max_c,
                                            int max_g, bool by_center, int *nk,
                                            int *nc, int *ng) {
  *nk = min(c[0], max_k);
  *nc = min(c[1], max_c);
  *ng = min(c[2], max_g);
  return 7 * *nk + 3 * *nc + (by_center ? 2 * *ng : 0) + *ng;
}

__global__ __launch_bounds__(kBlock) void decode_pack_kernel(
    ganet_decode_out o, int B, int max_k, int max_c, int max_g, int by_center,
    int32_t *__restrict__ packed, long long packed_words) {
  __shared__ int wave_sums[kWavesPerBlock];
  const int b = blockIdx.x, tid = threadIdx.x;
  int nk, nc, ng;
  int part = 0;
  for (int i = tid; i < b; i += kBlock)
    part += record_words(o.counts + (size_t)i * GANET_DECODE_COUNT_STRIDE, max_k, max_c,
                         max_g, by_center, &nk, &nc, &ng);
  int off;
  block_excl_scan(part, wave_sums, &off);  // off = total of the partial sums
  const int32_t *c = o.counts + (size_t)b * GANET_DECODE_COUNT_STRIDE;
  const int words = record_words(c, max_k, max_c, max_g, by_center, &nk, &nc, &ng);
  const int ngc = by_center ? ng : 0;
  int32_t *hdr = packed + (size_t)b * 8;
  const long long rec0 = (long long)B * 8 + off;
  const bool fits = rec0 + words <= packed_words;
  if (tid == 0) {
    hdr[0] = c[0];
    hdr[1] = c[1];
    hdr[2] = c[2];
    hdr[3] = c[3] | (fits ? 0 : 8);
    hdr[4] = off;
    hdr[5] = words;
    if (b != 0) hdr[6] = 0;
    hdr[7] = 0;
    if (b == B - 1) packed[6] = off + words;
  }
  if (!fits) return;
  int32_t *r = packed + rec0;
  const int32_t *kxy = o.kpt_xy + (size_t)b * max_k * 2;
  const int32_t *al = reinterpret_cast<const int32_t *>(o.kpt_align) + (size_t)b * max_k * 2;
  const int32_t *rt = reinterpret_cast<const int32_t *>(o.kpt_root) + (size_t)b * max_k * 2;
  const int32_t *kg = o.kpt_gid + (size_t)b * max_k;
  const int32_t *cxy = o.centers_xy + (size_t)b * max_c * 2;
  const int32_t *cg = o.center_gid + (size_t)b * max_c;
  const int32_t *gc = o.group_center + (size_t)b * max_g * 2;
  const int32_t *gn = o.group_count + (size_t)b * max_g;
  for (int i = tid; i < 2 * nk; i += kBlock) {
    r[i] = kxy[i];
    r[2 * nk + i] = al[i];
    r[4 * nk + i] = rt[i];
  }
  r += 6 * nk;
  for (int i = tid; i < nk; i += kBlock) r[i] = kg[i];
  r += nk;
  for (int i = tid; i < 2 * nc; i += kBlock) r[i] = cxy[i];
  r += 2 * nc;
  for (int i = tid; i < nc; i += kBlock) r[i] = cg[i];
  r += nc;
  for (int i = tid; i < 2 * ngc; i += kBlock) r[i] = gc[i];
  r += 2 * ngc;
  for (int i = tid; i < ng; i += kBlock) r[i] = gn[i];
}

}  // namespace

#ifdef GANET_DECODE_STAMPS
extern "C" int ganet_debug_decode_stamps(unsigned long long *host16) {
  return hipMemcpyFromSymbol(host16, HIP_SYMBOL(g_decode_stamps), 16 * sizeof(unsigned long long)) == hipSuccess
             ? GANET_OK : GANET_ERR_HIP;
}
#endif

extern "C" int ganet_decode_pack(const ganet_decode_params *p,
                                 const ganet_decode_out *out, int32_t *packed,
                                 size_t packed_words, ganet_stream_t stream) {
  if (!p || !out || !packed) return GANET_ERR_NULL;
  if (!out->counts || !out->centers_xy || !out->center_gid || !out->kpt_xy ||
      !out->kpt_align || !out->kpt_root || !out->kpt_gid || !out->group_center ||
      !out->group_count)
    return GANET_ERR_NULL;
  if (p->B <= 0 || p->max_kpts <= 0 || p->max_centers <= 0 || p->max_groups <= 0)
    return GANET_ERR_BAD_SHAPE;
  if (packed_words < (size_t)p->B * 8) return GANET_ERR_WORKSPACE;
  hipLaunchKernelGGL(decode_pack_kernel, dim3(p->B), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), *out, p->B, p->max_kpts,
                     p->max_centers, p->max_groups, p->by_center_thr >= 0 ? 1 : 0,
                     packed, (long long)packed_words);
  return ganet_check_launch();
}

extern "C" size_t ganet_decode_workspace_bytes(const ganet_decode_params *p) {
  if (!p || p->B <= 0 || p->H <= 0 || p->W <= 0) return 0;
  const size_t nwords = ((size_t)p->H * p->W + 63) / 64;
  return ganet_align_up((size_t)p->B * nwords * sizeof(unsigned long long), 256);
}

namespace {

// shared by ganet_decode_batch (bitmap in the caller's workspace) and
// ganet_decode_batch_noalloc (bitmap in the library's static device array)
int decode_batch_check(const float *heat, const float *offset, const float *error,
                       const ganet_decode_params *p, const ganet_decode_out *out) {
  if (!heat || !offset || !error || !p || !out) return GANET_ERR_NULL;
  if (!out->counts || !out->centers_xy || !out->center_gid || !out->kpt_xy ||
      !out->kpt_align || !out->kpt_root || !out->kpt_gid ||
      !out->group_center || !out->group_count)
    return GANET_ERR_NULL;
  if (p->B <= 0 || p->H <= 0 || p->W <= 0 || p->max_kpts <= 0 ||
      p->max_centers <= 0 || p->max_groups <= 0)
    return GANET_ERR_BAD_SHAPE;
  if (p->H > 65535 || p->W > 65535 ||
      (long long)p->H * p->W > (long long)INT32_MAX - 4096)
    return GANET_ERR_UNSUPPORTED;
  if (p->max_centers > kMaxCenters) return GANET_ERR_UNSUPPORTED;
  if (p->B > 65535) return GANET_ERR_UNSUPPORTED;  // grid.y of the mask kernel
  return GANET_OK;
}

// `packed` != nullptr asks for the record stream as well (ganet_decode_batch_packed): written
// by the group kernel itself where one workgroup per image does everything, by
// decode_pack_kernel behind the grid-wide kernels of large by-centre images.
int decode_batch_launch(const float *heat, const float *offset, const float *error,
                        const ganet_decode_params *p, const ganet_decode_out *out,
                        unsigned long long *kmask, hipStream_t s, int32_t *packed = nullptr,
                        size_t packed_words = 0, unsigned long long *pack_ctr = nullptr) {
  const int HW = p->H * p->W;
  const int nwords = (HW + 63) / 64;

  const bool fused_mask = nwords <= kFusedMaskWords;
  // Large images with by-centre grouping (or decode only): keypoint lists and keypoint
  // assignment on grids over the batch, only the order-dependent centre clustering stays
  // one workgroup per image.  The no-centre mode clusters the keypoints themselves in
  // order and keeps the one-workgroup form.
  const bool split = !fused_mask && (p->by_center_thr >= 0 || (p->flags & 1));
  if (!fused_mask) {
    dim3 g1((HW + kBlock - 1) / kBlock, p->B);
    hipLaunchKernelGGL(decode_mask_kernel, g1, dim3(kBlock), 0, s, heat, HW,
                       p->W, p->kpt_thr, kmask, nwords, p->flags & 2, p->flags & 8,
                       split ? out->counts : nullptr);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
  }

  DecodeArgs a;
  a.heat = heat;
  a.offset = offset;
  a.error = error;
  a.kmask = fused_mask ? nullptr : kmask;
  a.kpt_thr = p->kpt_thr;
  a.H = p->H;
  a.W = p->W;
  a.nwords = nwords;
  a.root_thr = p->root_thr;
  a.cluster_thr = p->cluster_thr;
  a.by_center_thr = p->by_center_thr;
  a.cluster_sq = sqrt_le_bound(p->cluster_thr);
  a.by_center_sq = sqrt_le_bound(p->by_center_thr);
  a.max_kpts = p->max_kpts;
  a.max_centers = p->max_centers;
  a.max_groups = p->max_groups;
  a.flags = p->flags;
  a.out = *out;
  a.packed = nullptr;
  a.packed_words = 0;
  a.pack_ctr = nullptr;
  a.B = p->B;
  auto pack_behind = [&]() {
    hipLaunchKernelGGL(decode_pack_kernel, dim3(p->B), dim3(kBlock), 0, s, *out, p->B,
                       p->max_kpts, p->max_centers, p->max_groups,
                       p->by_center_thr >= 0 ? 1 : 0, packed, (long long)packed_words);
    return ganet_check_launch();
  };
  if (split) {
    dim3 gg((nwords + kBlock - 1) / kBlock, p->B);
    hipLaunchKernelGGL(decode_gather_kernel, gg, dim3(kBlock), 0, s, a);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    hipLaunchKernelGGL(decode_group_kernel<2>, dim3(p->B), dim3(kBlock), 0, s, a);
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    if (!(p->flags & 1)) {
      dim3 ga((p->max_kpts + kBlock - 1) / kBlock, p->B);
      hipLaunchKernelGGL(decode_assign_kernel, ga, dim3(kBlock), 0, s, a);
    }
    if (ganet_check_launch() != GANET_OK) return GANET_ERR_HIP;
    return packed ? pack_behind() : GANET_OK;
  }
  if (packed) {
    a.packed = packed;
    a.packed_words = (long long)(packed_words < (size_t)INT32_MAX ? packed_words : INT32_MAX);
    a.pack_ctr = pack_ctr;
  }
  hipLaunchKernelGGL(decode_group_kernel<0>, dim3(p->B), dim3(kBlock), 0, s, a);
  return ganet_check_launch();
}

// ---- the keypoint bitmap of ganet_decode_batch_noalloc --------------------------------
// One copy per GPU, held until the process ends.  `g_decode_scratch` is a host-side
// stand-in, not a device address: host code only ever uses the per-device address that
// hipGetSymbolAddress returned (decode_scratch_ptr).
constexpr size_t kStaticScratchBytes = GANET_DECODE_NOALLOC_MAX_BYTES;
static __device__ __align__(256) unsigned long long
    g_decode_scratch[kStaticScratchBytes / sizeof(unsigned long long)];

constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
unsigned long long *g_scratch_ptr[kMaxDevices] = {};

// address of the current device's copy of g_decode_scratch (looked up once per device)
int decode_scratch_ptr(unsigned long long **ptr) {
  int dev = -1;
  GANET_RETURN_IF_HIP(hipGetDevice(&dev));
  if (dev < 0 || dev >= kMaxDevices) return GANET_ERR_UNSUPPORTED;
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (!g_scratch_ptr[dev]) {
    void *q = nullptr;
    GANET_RETURN_IF_HIP(hipGetSymbolAddress(&q, HIP_SYMBOL(g_decode_scratch)));
    if (!q) return GANET_ERR_HIP;
    g_scratch_ptr[dev] = static_cast<unsigned long long *>(q);
  }
  *ptr = g_scratch_ptr[dev];
  return GANET_OK;
}

}  // namespace

extern "C" int ganet_decode_batch(const float *heat, const float *offset,
                                  const float *error,
                                  const ganet_decode_params *p,
                                  const ganet_decode_out *out, void *workspace,
                                  size_t workspace_bytes,
                                  ganet_stream_t stream) {
  if (!workspace) return GANET_ERR_NULL;
  const int rc = decode_batch_check(heat, offset, error, p, out);
  if (rc != GANET_OK) return rc;
  if (workspace_bytes < ganet_decode_workspace_bytes(p))
    return GANET_ERR_WORKSPACE;
  return decode_batch_launch(heat, offset, error, p, out,
                             static_cast<unsigned long long *>(workspace),
                             static_cast<hipStream_t>(stream));
}

extern "C" size_t ganet_decode_packed_workspace_bytes(const ganet_decode_params *p) {
  const size_t n = ganet_decode_workspace_bytes(p);
  return n ? n + 256 : 0;   // + the two counters of the record stream, on a line of their own
}

extern "C" int ganet_decode_batch_packed(const float *heat, const float *offset,
                                         const float *error, const ganet_decode_params *p,
                                         const ganet_decode_out *out, int32_t *packed,
                                         size_t packed_words, void *workspace,
                                         size_t workspace_bytes, ganet_stream_t stream) {
  if (!workspace || !packed) return GANET_ERR_NULL;
  const int rc = decode_batch_check(heat, offset, error, p, out);
  if (rc != GANET_OK) return rc;
  if (packed_words < (size_t)p->B * 8) return GANET_ERR_WORKSPACE;
  if (workspace_bytes < ganet_decode_packed_workspace_bytes(p)) return GANET_ERR_WORKSPACE;
  unsigned char *ws = static_cast<unsigned char *>(workspace);
  return decode_batch_launch(
      heat, offset, error, p, out, reinterpret_cast<unsigned long long *>(ws),
      static_cast<hipStream_t>(stream), packed, packed_words,
      reinterpret_cast<unsigned long long *>(ws + ganet_decode_workspace_bytes(p)));
}

extern "C" int ganet_decode_batch_noalloc(const float *heat, const float *offset,
                                          const float *error,
                                          const ganet_decode_params *p,
                                          const ganet_decode_out *out,
                                          ganet_stream_t stream) {
  const int rc = decode_batch_check(heat, offset, error, p, out);
  if (rc != GANET_OK) return rc;
  // the bitmap must fit the static array: B * ceil(H*W/64) words
  if (ganet_decode_workspace_bytes(p) > kStaticScratchBytes) return GANET_ERR_WORKSPACE;
  unsigned long long *kmask = nullptr;
  const int rp = decode_scratch_ptr(&kmask);
  if (rp != GANET_OK) return rp;
  // every word of the bitmap the group kernel reads is written by the mask kernel of the
  // same call, so nothing has to start at zero
  return decode_batch_launch(heat, offset, error, p, out, kmask,
                             static_cast<hipStream_t>(stream));
}

extern "C" int ganet_group_batch(const ganet_decode_params *p,
                                 const ganet_decode_out *out,
                                 ganet_stream_t stream) {
  if (!p || !out) return GANET_ERR_NULL;
  if (!out->counts || !out->centers_xy || !out->center_gid || !out->kpt_root ||
      !out->kpt_gid || !out->group_center || !out->group_count)
    return GANET_ERR_NULL;
  if (p->B <= 0 || p->max_kpts <= 0 || p->max_centers <= 0 || p->max_groups <= 0)
    return GANET_ERR_BAD_SHAPE;
  if (p->max_centers > kMaxCenters) return GANET_ERR_UNSUPPORTED;
  DecodeArgs a = {};
  a.H = p->H;
  a.W = p->W;
  a.cluster_thr = p->cluster_thr;
  a.by_center_thr = p->by_center_thr;
  a.cluster_sq = sqrt_le_bound(p->cluster_thr);
  a.by_center_sq = sqrt_le_bound(p->by_center_thr);
  a.max_kpts = p->max_kpts;
  a.max_centers = p->max_centers;
  a.max_groups = p->max_groups;
  a.flags = p->flags & ~1;
  a.out = *out;
  hipLaunchKernelGGL(decode_group_kernel<1>, dim3(p->B), dim3(kBlock), 0,
                     static_cast<hipStream_t>(stream), a);
  return ganet_check_launch();
}
