// Deformable convolution forward, contract-then-sample form as ONE persistent launch
// ("algorithm 3") for gfx950 (MI355X).
//
// The two kernels of csrc/deform_conv_cts.hip (per-tap GEMM  Y_t = X * W_t^T, then
// out = sum_t sample(Y_t)) replace deformable_im2col + addmm_
// (mmdet/ops/dcn/src/cuda/deform_conv_cuda.cpp:222-236) with 2 launches per chunk of the
// batch, 8 per LFA call at 128 images: every launch ramps up and drains on its own, the
// matrix cores idle while the sampling kernel runs and the vector units idle during the
// GEMM.  Here one workgroup per CU stays resident for the whole call; its first waves (one
// per SIMD) produce Y, the others sample it:
//
//   produce   one item = 128 pixels of one image x the CU's share of the T*Cout columns of Y.
//             The share's weights sit in LDS for the whole launch (no slice staging, no
//             barrier): with T*Cout*Cin*2 bytes over the LDS size the columns are split
//             over `nsplit` kinds of workgroup (224 of 448 columns each at the LFA shape).
//             x goes from memory straight into MFMA operands, the weight rows are read
//             through a permutation that leaves 16 consecutive outputs of one pixel in
//             every lane's accumulator; two 32-column blocks are parked in a wave-private
//             LDS tile and leave as whole 128-byte lines.
//   consume   one item = 64 output pixels: the pixel-group loop of cts_sample_kernel
//             (8 lanes x 16 B per pixel, 4 buffer loads per lane and tap, a corner outside
//             the map reads 0 through the buffer bounds check).
//
// Flow: the images are dealt to 8 queues; a queue is served by the workgroups of ONE XCD
// (the one whose id, read from the hardware register, took the queue's owner word), so Y is
// written and read through one L2 and needs no write-back or invalidate in between: a
// producing wave waits for its own stores (s_waitcnt vmcnt(0): they have reached L2) and
// adds to the counter of the item's band of 256 pixels; a consuming wave forms the corners
// of its 8 pixels first, knows from them the last band it reads, and loads only after every
// band up to that one has its full count.  No line of Y is loaded before it is complete, so
// no L1 holds an old copy of it within the launch.  Both kinds of item are claimed in order
// from counters.  Producing never waits and every resident workgroup has producing waves, so
// every band a sampling wave waits for gets done whatever the placement; a producing wave
// samples once no item of its kind is left.
// The hand-off is NOT a release/acquire pair: the counter add and its poll are relaxed, and
// what orders Y behind the flag is (a) the producer's s_waitcnt vmcnt(0) (its stores have been
// acknowledged by the L2 both sides share), (b) the consumer issuing its loads of Y after the
// poll returned (a compiler barrier keeps the order; the hardware returns a wave's loads in
// order) and (c) no L1 holding an old copy (no line is loaded before it is complete).  A change
// of any of the three needs the agent-scope forms; tests/test_gpu_dcn_algo3.py's bit-identity
// with algorithm 2 is the gate.
// Nothing depends on dispatch order.  What the launch needs is an XCD that holds workgroups
// of every kind at once: owner words are only taken by such an XCD.  Every wait is bounded
// (STREAM_TIMEOUT_TICKS); a timeout sets ctl[CTL_ERROR] and every wave leaves.
// The control block is zeroed by a memset node ahead of every launch.
//
// Arithmetic of the sampling position and weights: deform_conv_cuda_kernel.cu:84-115,
// 218-229 (and :604-627 for the DCNv2 mask), unchanged.

#include "dcn_device.h"
#include <cstdlib>

namespace ganet_dcn {

namespace {

#ifndef GANET_STREAM_WAVES
#define GANET_STREAM_WAVES 16
#endif
constexpr int FW = GANET_STREAM_WAVES;   // waves per workgroup, one workgroup per CU
constexpr int FNT = FW * 64;
constexpr int PITEM = 128;               // pixels per producer item
constexpr int CITEM = 64;                // pixels per consumer item
constexpr int CSTEP = 8;                 // pixels per consumer step (8 lanes each)
constexpr int BANDPX = 256;              // pixels per completion counter
constexpr int NQ = 8;                    // queues (one per XCD)
constexpr int SMEM_BYTES = 160 * 1024 - 256;
constexpr int STAGE_ROW = 128 + 16;      // bytes per pixel row of a producing wave's LDS tile
constexpr int STAGE_BYTES = 32 * STAGE_ROW;
#ifndef GANET_STREAM_PROD_WAVES
#define GANET_STREAM_PROD_WAVES 4
#endif
constexpr int STREAM_PROD_WAVES = GANET_STREAM_PROD_WAVES;   // producing waves per workgroup
constexpr unsigned long long STREAM_TIMEOUT_TICKS = 200000000ull;  // 2 s of the 100 MHz clock

static_assert(BANDPX % PITEM == 0, "a producer item lies in one band");

// control block (32-bit words); hot words sit in 128-byte lines of their own
enum {
  CTL_LINE = 32,
  CTL_ERROR = 0,
  CTL_STAMPS = 8,                // GANET_STREAM_DEBUG builds: 8 x 64-bit phase times
  CTL_OWNER = CTL_LINE,          // [NQ]  0 = free, else XCC id + 1
  CTL_JOINED = 2 * CTL_LINE,     // [8]   workgroups that joined, per XCC id
  CTL_QUEUE = 3 * CTL_LINE,      // per queue: producer claims of each kind, consumer claims
  Q_PROD = 0,                    // + kind * CTL_LINE
  Q_CONS = 4 * CTL_LINE,
  Q_WORDS = 5 * CTL_LINE,
  CTL_FIXED = CTL_QUEUE + NQ * Q_WORDS   // band counters follow: [image][band]
};

struct StreamArgs {
  const void *x;        // [B][H*W][Cp] channel-last, arithmetic dtype
  const void *wq;       // [T*CoutP][Cp]
  void *y;              // [B][H*W][T*CoutP]
  const float *offset;  // [B][2T][H][W]
  const float *mask;    // [B][T][H][W] or nullptr
  const void *bias;     // [Cout] or nullptr
  void *out;
  unsigned int *ctl;    // stream_ctl_bytes() words, zeroed before the launch
  int B, H, W, Cp, Cout, CoutP, T;
  int kW, pH, pW, dH, dW;
  int out_nhwc, relu, points;
  int nsplit, ncols;    // kinds of workgroup, columns of Y per kind
  int pipi, cipi;       // producer / consumer items per image
  int nbands;           // bands per image
  int prod_waves;       // waves 0..prod_waves-1 produce
  int whole_image;      // debug: consumers wait for the whole image
  int debug;            // timing experiments: bit0 produce nothing, bit1 sample nothing, bit2 no readiness wait
};

typedef __attribute__((address_space(1))) unsigned int gu32;

// GANET_STREAM_DEBUG builds: shader-clock time per phase, summed over waves into
// ctl[CTL_STAMPS + 2 * phase] (64 bit).  No output value depends on a stamp.
#ifdef GANET_STREAM_DEBUG
#define STAMP_DECL unsigned long long st_t = __builtin_amdgcn_s_memtime(), st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0}
#define STAMP(ph)                                                   \
  do {                                                              \
    const unsigned long long st_n = __builtin_amdgcn_s_memtime();   \
    st_acc[ph] += st_n - st_t;                                      \
    st_t = st_n;                                                    \
  } while (0)
#define STAMP_ARGS , unsigned long long &st_t, unsigned long long (&st_acc)[8]
#define STAMP_PASS , st_t, st_acc
#else
#define STAMP_DECL
#define STAMP(ph) do {} while (0)
#define STAMP_ARGS
#define STAMP_PASS
#endif

__device__ __forceinline__ unsigned int flag_load(const unsigned int *p) {
  return __hip_atomic_load((const gu32 *)p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int flag_add(unsigned int *p, unsigned int v) {
  return __hip_atomic_fetch_add((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ void flag_store(unsigned int *p, unsigned int v) {
  __hip_atomic_store((gu32 *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned int flag_cas(unsigned int *p, unsigned int expect,
                                                 unsigned int v) {
  __hip_atomic_compare_exchange_strong((gu32 *)p, &expect, v, __ATOMIC_RELAXED,
                                       __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  return expect;   // the value found
}

__device__ __forceinline__ void stg128(void *p, const u32x4 &v) {
  *(GANET_GLOBAL u32x4 *)(p) = v;
}

__device__ __forceinline__ void wave_sync_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// producer items a band of `px` pixels counts when complete
__device__ __forceinline__ unsigned band_target(const StreamArgs &a, int band) {
  const int px = min(BANDPX, a.H * a.W - band * BANDPX);
  return (unsigned)(a.nsplit * ((px + PITEM - 1) / PITEM));
}

// ---- produce: PITEM pixels of image `img`, the workgroup's ncols columns ----------------
// MFMA rows = outputs (weights from LDS), columns = pixels (x from memory).  Lane i reads
// weight row perm(i) = 16*((i>>2)&1) + 4*(i>>3) + (i&3) of the 32-column block: the
// accumulator registers 4g+e of lane (i, h) then hold outputs 16h + 4g + e of pixel i,
// 32 contiguous bytes of Y.  Two blocks (64 columns = one 128-byte line per pixel) are parked
// in the wave's LDS tile and stored as whole lines, 8 lanes x 16 B each.
template <typename TA, int KH>
__device__ __forceinline__ void produce_item(const StreamArgs &a, int img, int item, int kind,
                                             const unsigned char *s_wt, unsigned char *stage,
                                             int lane STAMP_ARGS) {
  constexpr int EPL = 8;
  constexpr int NQF = KH / 16;
  constexpr int LDB = KH + EPL;               // +16 B: conflict-free ds_read_b128
  typedef typename Mma<TA>::frag frag_t;
  const int i = lane & 31, h = lane >> 5;
  const int HW = a.H * a.W;
  const int N = a.T * a.CoutP;
  const int ncb = a.ncols / 32;
  const int perm = 16 * ((i >> 2) & 1) + 4 * (i >> 3) + (i & 3);
  const TA *wrow = reinterpret_cast<const TA *>(s_wt) + perm * LDB + h * EPL;
  const TA *x = static_cast<const TA *>(a.x) + (size_t)img * HW * a.Cp;
  unsigned char *y = static_cast<unsigned char *>(a.y) + (size_t)img * HW * N * 2 +
                     (size_t)kind * a.ncols * 2;
  unsigned char *st_w = stage + i * STAGE_ROW + h * 32;

#pragma unroll 1
  for (int pb = 0; pb < PITEM / 32; ++pb) {
    const int m0 = item * PITEM + pb * 32;
    if (m0 >= HW || (a.debug & 1)) break;
    const int r = m0 + i;
    const TA *xrow = x + (size_t)(r < HW ? r : HW - 1) * a.Cp + h * EPL;
    frag_t fa[NQF];
#pragma unroll
    for (int q = 0; q < NQF; ++q)
      fa[q] = __builtin_bit_cast(frag_t, ldg128(xrow + q * 2 * EPL));
#ifdef GANET_STREAM_DEBUG
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    STAMP(1);
#endif
#pragma unroll 1
    for (int cb = 0; cb < ncb; ++cb) {
      f32x16 acc;
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[e] = 0.f;
      const TA *wb = wrow + cb * 32 * LDB;
#pragma unroll
      for (int q = 0; q < NQF; ++q) {
        const frag_t fb = *reinterpret_cast<const frag_t *>(wb + q * 2 * EPL);
        Mma<TA>::mma(fb, fa[q], acc);
      }
      TA v[16];
#pragma unroll
      for (int e = 0; e < 16; ++e) v[e] = from_f32<TA>(acc[e]);
      u32x4 lo, hi;
      __builtin_memcpy(&lo, v, 16);
      __builtin_memcpy(&hi, v + 8, 16);
      *reinterpret_cast<u32x4 *>(st_w + (cb & 1) * 64) = lo;
      *reinterpret_cast<u32x4 *>(st_w + (cb & 1) * 64 + 16) = hi;
      if (cb & 1) {
        wave_sync_lds();
        const int pc = lane & 7, rq = lane >> 3;
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) {
          const int rr = ps * 8 + rq;
          const u32x4 t = *reinterpret_cast<const u32x4 *>(stage + rr * STAGE_ROW + pc * 16);
          if (m0 + rr < HW) stg128(y + ((size_t)(m0 + rr) * N + (cb - 1) * 32) * 2 + pc * 16, t);
        }
        wave_sync_lds();
      } else if (cb == ncb - 1) {   // an odd block count ends on half a line per pixel
        wave_sync_lds();
        const int pc = lane & 3, rq = lane >> 2;
#pragma unroll
        for (int ps = 0; ps < 2; ++ps) {
          const int rr = ps * 16 + rq;
          const u32x4 t = *reinterpret_cast<const u32x4 *>(stage + rr * STAGE_ROW + pc * 16);
          if (m0 + rr < HW) stg128(y + ((size_t)(m0 + rr) * N + cb * 32) * 2 + pc * 16, t);
        }
        wave_sync_lds();
      }
    }
    STAMP(2);
  }
  STAMP(2);
  // this wave's stores have reached L2; then count the item in its band
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (lane == 0) {
    unsigned int *cnt = a.ctl + CTL_FIXED;
    flag_add(&cnt[(size_t)img * a.nbands + (item * PITEM) / BANDPX], 1u);
  }
  STAMP(3);
}

// one 16-byte piece of a corner row, weight w, into fp32 accumulators
template <typename TY>
__device__ __forceinline__ void sacc_piece(const u32x4 &v, float w, f32x2 *acc);
template <>
__device__ __forceinline__ void sacc_piece<bf16_t>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const f32x2 d = {__uint_as_float(v[j] << 16), __uint_as_float(v[j] & 0xffff0000u)};
    acc[j] = d * w + acc[j];
  }
}
template <>
__device__ __forceinline__ void sacc_piece<f16_t>(const u32x4 &v, float w, f32x2 *acc) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const unsigned int u = v[j];
    const _Float16 lo = __builtin_bit_cast(_Float16, (unsigned short)(u & 0xffffu));
    const _Float16 hi = __builtin_bit_cast(_Float16, (unsigned short)(u >> 16));
    const f32x2 d = {(float)lo, (float)hi};
    acc[j] = d * w + acc[j];
  }
}

// ---- consume, first half: corners and weights of CSTEP pixels into the wave's LDS table.
// Lane (pj, piece) forms taps piece, piece + 8, ... of pixel pj.  Returns the last band of Y
// this lane's corners touch (-1: none).
__device__ __forceinline__ int consume_corners(const StreamArgs &a, int img, int p0, int lane,
                                               unsigned int *s_off, float *s_w) {
  const int pj = lane >> 3, piece = lane & 7;
  const int HW = a.H * a.W;
  const int N = a.T * a.CoutP;
  const unsigned px_bytes = (unsigned)N * 2;
  const unsigned row_bytes = (unsigned)a.W * px_bytes;
  const int p = p0 + pj;
  const bool valid = p < HW;
  const int ho = valid ? p / a.W : 0;
  const int wo = valid ? p - ho * a.W : 0;
  int last = -1;
  for (int t = piece; t < a.T; t += 8) {
    unsigned c1 = OOB_OFFSET, c2 = OOB_OFFSET, c3 = OOB_OFFSET, c4 = OOB_OFFSET;
    float w1 = 0.f, w2 = 0.f, w3 = 0.f, w4 = 0.f;
    if (valid) {
      const float *op = a.offset + ((size_t)img * 2 * a.T + 2 * t) * HW + p;
      const float off_h = ldg_f32(op);
      float off_w = ldg_f32(op + HW);
      const float mk = a.mask ? ldg_f32(a.mask + ((size_t)img * a.T + t) * HW + p) : 1.f;
      const int ti = t / a.kW, tj = t - ti * a.kW;
      if (a.points)  // dcn_offset = pts - base  (lanepoints_conv.py:63-74,170)
        off_w = off_w - (float)(tj - a.pW);
      const float h_im = (float)(ho - a.pH + ti * a.dH) + off_h;
      const float w_im = (float)(wo - a.pW + tj * a.dW) + off_w;
      if (h_im > -1 && w_im > -1 && h_im < a.H && w_im < a.W) {
        const int h_low = (int)floorf(h_im), w_low = (int)floorf(w_im);
        const int h_high = h_low + 1, w_high = w_low + 1;
        const float lh = h_im - (float)h_low, lw = w_im - (float)w_low;
        const float hh = 1.f - lh, hw_ = 1.f - lw;
        w1 = hh * hw_ * mk;
        w2 = hh * lw * mk;
        w3 = lh * hw_ * mk;
        w4 = lh * lw * mk;
        const int lin = h_low * a.W + w_low;
        const unsigned base = (unsigned)h_low * row_bytes + (unsigned)w_low * px_bytes +
                              (unsigned)t * (unsigned)(a.CoutP * 2);
        const bool hl = h_low >= 0, hh_ok = h_high <= a.H - 1;
        const bool wl = w_low >= 0, wh_ok = w_high <= a.W - 1;
        if (hl && wl) { c1 = base; last = max(last, lin); }
        if (hl && wh_ok) { c2 = base + px_bytes; last = max(last, lin + 1); }
        if (hh_ok && wl) { c3 = base + row_bytes; last = max(last, lin + a.W); }
        if (hh_ok && wh_ok) { c4 = base + row_bytes + px_bytes; last = max(last, lin + a.W + 1); }
      }
    }
    *reinterpret_cast<uint4 *>(s_off + (pj * a.T + t) * 4) = make_uint4(c1, c2, c3, c4);
    *reinterpret_cast<float4 *>(s_w + (pj * a.T + t) * 4) = make_float4(w1, w2, w3, w4);
  }
  wave_sync_lds();
  return last < 0 ? -1 : last / BANDPX;
}

// ---- consume, second half: gather, blend, store ------------------------------------------
template <typename TY, typename TO>
__device__ __forceinline__ void consume_sample(const StreamArgs &a, int img, int p0, int lane,
                                               const unsigned int *s_off, const float *s_w) {
  constexpr int EPL = 8, NACC = 4;
  const int pj = lane >> 3, piece = lane & 7;
  const int HW = a.H * a.W;
  const int N = a.T * a.CoutP;
  const size_t img_bytes = (size_t)HW * N * 2;
  const __amdgpu_buffer_rsrc_t ysrc = __builtin_amdgcn_make_buffer_rsrc(
      static_cast<unsigned char *>(a.y) + (size_t)img * img_bytes, 0, (int)img_bytes, 0x00020000);
  const unsigned lane_off = (unsigned)(piece * EPL * 2);
  const int p = p0 + pj;

  f32x2 acc[NACC];
#pragma unroll
  for (int j = 0; j < NACC; ++j) acc[j] = f32x2{0.f, 0.f};
  const unsigned int *to = s_off + pj * a.T * 4;
  const float *tw = s_w + pj * a.T * 4;
  for (int t0 = 0; t0 < a.T; t0 += 4) {
    // the loads of four taps (16 x 16 B per lane) go out before the first one is used
    u32x4 v[4][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (t0 + j < a.T) {
        const uint4 cx = *reinterpret_cast<const uint4 *>(to + (t0 + j) * 4);
        v[j][0] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.x + lane_off, 0, 0);
        v[j][1] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.y + lane_off, 0, 0);
        v[j][2] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.z + lane_off, 0, 0);
        v[j][3] = __builtin_amdgcn_raw_buffer_load_b128(ysrc, cx.w + lane_off, 0, 0);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (t0 + j < a.T) {
        const float4 w = *reinterpret_cast<const float4 *>(tw + (t0 + j) * 4);
        sacc_piece<TY>(v[j][0], w.x, acc);
        sacc_piece<TY>(v[j][1], w.y, acc);
        sacc_piece<TY>(v[j][2], w.z, acc);
        sacc_piece<TY>(v[j][3], w.w, acc);
      }
    }
  }
  wave_sync_lds();   // the table is free again

  if (p < HW) {
    const TO *bias = static_cast<const TO *>(a.bias);
    TO *out = static_cast<TO *>(a.out);
    const int c0 = piece * EPL;
    float r[EPL];
#pragma unroll
    for (int j = 0; j < NACC; ++j) {
      r[2 * j] = acc[j].x;
      r[2 * j + 1] = acc[j].y;
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
      if (bias && c0 + e < a.Cout) r[e] += to_f32(bias[c0 + e]);
      if (a.relu) r[e] = fmaxf(r[e], 0.f);
    }
    const size_t pg = (size_t)img * HW + p;   // pixel index in the whole batch
    if (a.out_nhwc) {
      TO *dst = out + pg * a.Cout + c0;
      if (c0 + EPL <= a.Cout && (a.Cout * sizeof(TO)) % 16 == 0 && sizeof(TO) == 2) {
        u32x4 pk;
        TO tmp[EPL];
#pragma unroll
        for (int e = 0; e < EPL; ++e) tmp[e] = from_f32<TO>(r[e]);
        __builtin_memcpy(&pk, tmp, 16);
        stg128(dst, pk);
      } else {
#pragma unroll
        for (int e = 0; e < EPL; ++e)
          if (c0 + e < a.Cout) dst[e] = from_f32<TO>(r[e]);
      }
    } else {
#pragma unroll
      for (int e = 0; e < EPL; ++e)
        if (c0 + e < a.Cout)
          out[((size_t)img * a.Cout + c0 + e) * HW + p] = from_f32<TO>(r[e]);
    }
  }
}

__device__ __forceinline__ int queue_images(int B, int q) { return q < B ? (B - q + NQ - 1) / NQ : 0; }

// one lane: a queue this XCD serves that still has unclaimed items of the wave's role
// (producer items of its kind, or consumer items), or -1 when there is none left (or the
// launch has failed).
__device__ int find_queue(const StreamArgs &a, unsigned xcc, int kind, bool producing) {
  unsigned int *ctl = a.ctl;
  const unsigned me = xcc + 1u;
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  for (unsigned spins = 0;; ++spins) {
    bool pending = false;
    for (int dq = 0; dq < NQ; ++dq) {
      const int q = (int)((xcc + dq) % NQ);
      const int nimg = queue_images(a.B, q);
      if (nimg == 0) continue;
      unsigned o = flag_load(&ctl[CTL_OWNER + q]);
      if (o == 0) {
        // only an XCD that holds every kind of workgroup can finish a queue
        if (flag_load(&ctl[CTL_JOINED + xcc]) >= (unsigned)a.nsplit)
          o = flag_cas(&ctl[CTL_OWNER + q], 0u, me), o = o == 0 ? me : o;
        else
          pending = true;
      }
      if (o != me) continue;
      const unsigned int *qw = ctl + CTL_QUEUE + q * Q_WORDS;
      if (producing ? flag_load(&qw[Q_PROD + kind * CTL_LINE]) < (unsigned)(nimg * a.pipi)
                    : flag_load(&qw[Q_CONS]) < (unsigned)(nimg * a.cipi))
        return q;
    }
    if (!pending) return -1;
    __builtin_amdgcn_s_sleep(64);
    if ((spins & 15) == 15) {
      if (flag_load(&ctl[CTL_ERROR]) != 0) return -1;
      if (__builtin_amdgcn_s_memrealtime() - t0 > STREAM_TIMEOUT_TICKS) {
        flag_store(&ctl[CTL_ERROR], 3u);
        return -1;
      }
    }
  }
}

template <typename TA, typename TO, int KH>
__global__ __launch_bounds__(FNT, FW / 4) void lfa_stream_kernel(StreamArgs a) {
  __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM_BYTES];
  __shared__ int s_info[2];
  constexpr int LDB = KH + 8;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
  unsigned int *ctl = a.ctl;
  unsigned int *cnt = ctl + CTL_FIXED;

  // join: the XCD this workgroup runs on, and its kind (which columns of Y it produces)
  if (tid == 0) {
    unsigned xid;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xid));
    xid &= 7u;
    const unsigned j = flag_add(&ctl[CTL_JOINED + xid], 1u);
    s_info[0] = (int)xid;
    s_info[1] = (int)(j % (unsigned)a.nsplit);
  }
  __syncthreads();
  const unsigned xcc = (unsigned)__builtin_amdgcn_readfirstlane(s_info[0]);
  const int kind = __builtin_amdgcn_readfirstlane(s_info[1]);

  // this kind's weight rows into LDS, once
  {
    const TA *wq = static_cast<const TA *>(a.wq) + (size_t)kind * a.ncols * a.Cp;
    TA *wt = reinterpret_cast<TA *>(smem);
    const int pieces = a.ncols * (KH / 8);
    for (int q = tid; q < pieces; q += FNT) {
      const int r = q / (KH / 8), pc = q - r * (KH / 8);
      *reinterpret_cast<u32x4 *>(wt + r * LDB + pc * 8) = ldg128(wq + (size_t)r * a.Cp + pc * 8);
    }
  }
  __syncthreads();
  // LDS beyond the weights: an output tile per producing wave (it doubles as that wave's
  // corner table once it samples), a corner table per sampling wave
  const int wt_bytes = a.ncols * LDB * 2;
  const int tab_bytes = CSTEP * a.T * 32;
  const int preg = tab_bytes > STAGE_BYTES ? tab_bytes : STAGE_BYTES;
  unsigned char *region = smem + wt_bytes +
                          (wid < a.prod_waves ? wid * preg
                                              : a.prod_waves * preg + (wid - a.prod_waves) * tab_bytes);
  unsigned int *s_off = reinterpret_cast<unsigned int *>(region);
  float *s_w = reinterpret_cast<float *>(s_off + CSTEP * a.T * 4);

  const int HW = a.H * a.W;
  bool producing = wid < a.prod_waves;
  int q = -1, ptotal = 0, ctotal = 0;
  unsigned int *qw = ctl;
  int wm_img = -1, wm = 0;
  bool failed = false;
  STAMP_DECL;

  while (!failed) {
    if (q < 0) {
      int nq = -1;
      if (lane == 0) nq = find_queue(a, xcc, kind, producing);
      q = __builtin_amdgcn_readfirstlane(nq);
      if (q < 0) {
        if (!producing) break;
        producing = false;   // no item of this kind is left anywhere: sample from now on
        continue;
      }
      const int nimg = queue_images(a.B, q);
      ptotal = nimg * a.pipi;
      ctotal = nimg * a.cipi;
      qw = ctl + CTL_QUEUE + q * Q_WORDS;
    }
    int k = -1;
    if (lane == 0) {
      const unsigned kk = flag_add(&qw[producing ? Q_PROD + kind * CTL_LINE : Q_CONS], 1u);
      k = kk < (unsigned)(producing ? ptotal : ctotal) ? (int)kk : -1;
    }
    k = __builtin_amdgcn_readfirstlane(k);
    if (k < 0) {
      q = -1;
      continue;
    }
    STAMP(0);
    if (producing) {
      produce_item<TA, KH>(a, q + NQ * (k / a.pipi), k % a.pipi, kind, smem, region, lane STAMP_PASS);
      continue;
    }

    // ---- a consumer item: CITEM pixels in steps of CSTEP
    const int c_img = q + NQ * (k / a.cipi);
    const int c_px = (k % a.cipi) * CITEM;
    if (wm_img != c_img) {
      wm_img = c_img;
      wm = 0;
    }
    for (int st = 0; st < CITEM / CSTEP && !failed; ++st) {
      const int p0 = c_px + st * CSTEP;
      if (p0 >= HW) break;
      int need = consume_corners(a, c_img, p0, lane, s_off, s_w);
      if (a.whole_image && need >= 0) need = a.nbands - 1;
      STAMP(4);
      if (!(a.debug & 4)) {
        unsigned spins = 0;
        unsigned long long t0 = 0;
        while (__any(need >= wm)) {
          // one round trip: the counters of bands wm .. wm + 63, one per lane
          const int b = wm + lane;
          bool open = false;
          if (b < a.nbands)
            open = flag_load(&cnt[(size_t)c_img * a.nbands + b]) < band_target(a, b);
          const unsigned long long m = __ballot(open);
          wm += m ? (int)__builtin_ctzll(m) : min(64, a.nbands - wm);
          if (!__any(need >= wm)) break;
          // the band is claimed by a producing wave or will be: those never wait
          if (spins == 0) t0 = __builtin_amdgcn_s_memrealtime();
          __builtin_amdgcn_s_sleep(32);
          if ((++spins & 63) == 0) {
            int stop = 0;
            if (lane == 0) {
              if (flag_load(&ctl[CTL_ERROR]) != 0) stop = 1;
              else if (__builtin_amdgcn_s_memrealtime() - t0 > STREAM_TIMEOUT_TICKS) {
                flag_store(&ctl[CTL_ERROR], 1u);
                stop = 1;
              }
            }
            if (__builtin_amdgcn_readfirstlane(stop)) failed = true;
            if (failed) break;
          }
        }
      }
      STAMP(5);
      if (failed) break;
      asm volatile("" ::: "memory");
      if (!(a.debug & 2)) consume_sample<TA, TO>(a, c_img, p0, lane, s_off, s_w);
#ifdef GANET_STREAM_DEBUG
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      STAMP(6);
#endif
    }
  }
#ifdef GANET_STREAM_DEBUG
  STAMP(0);
  if (lane == 0)
    for (int ph = 0; ph < 8; ++ph)
      __hip_atomic_fetch_add((__attribute__((address_space(1))) unsigned long long *)(ctl + CTL_STAMPS + 2 * ph),
                             st_acc[ph], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#endif
}

int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

// kinds of workgroup: the fewest whose share of the weights fits LDS beside the tables
int stream_nsplit(int T, int CoutP, int Cp) {
  const int N = T * CoutP;
  for (int s = 1; s <= 4; ++s) {
    if (N % (32 * s)) continue;
    const size_t tab = (size_t)CSTEP * T * 32;
    const size_t need = (size_t)(N / s) * (Cp + 8) * 2 +
                        STREAM_PROD_WAVES * (tab > STAGE_BYTES ? tab : (size_t)STAGE_BYTES) +
                        (FW - STREAM_PROD_WAVES) * tab;
    if (need <= (size_t)SMEM_BYTES) return s;
  }
  return 0;
}

size_t stream_img_bytes(const ganet_dcn_params *p, int CoutP) {
  return (size_t)p->H * p->W * p->kH * p->kW * CoutP * 2;
}

}  // namespace

// images whose Y one launch holds
int stream_chunk_images(const ganet_dcn_params *p, int CoutP) {
  const size_t per_img = stream_img_bytes(p, CoutP);
  static const size_t budget = (size_t)env_int("GANET_STREAM_Y_MB", 1024) << 20;   // once per process
  size_t n = budget / (per_img ? per_img : 1);
  if (n < (size_t)NQ) n = NQ;
  if (n > (size_t)p->B) n = p->B;
  return (int)n;
}

// 16-bit arithmetic, one K chunk (Cp in {64, 128, 256}), 64 output channels per tap, "same"
// output size, the weights of one kind of workgroup fit LDS
bool stream_applicable(const ganet_dcn_params *p, int ta, int Cp, int CoutP, int Ho, int Wo) {
  if (ta == GANET_F32) return false;
  if (Cp != 64 && Cp != 128 && Cp != 256) return false;
  if (CoutP != 64 || p->deformable_groups != 1 || p->groups != 1) return false;
  if (p->sH != 1 || p->sW != 1 || Ho != p->H || Wo != p->W) return false;
  if (p->kH * p->kW > 64) return false;
  if (stream_nsplit(p->kH * p->kW, CoutP, Cp) == 0) return false;
  return stream_img_bytes(p, CoutP) < 0x7fffffffull;
}

static size_t stream_ctl_bytes(const ganet_dcn_params *p, int nimg) {
  const size_t nbands = ((size_t)p->H * p->W + BANDPX - 1) / BANDPX;
  return ganet_align_up((CTL_FIXED + (size_t)nimg * nbands) * 4, 256);
}

size_t stream_workspace_bytes(const ganet_dcn_params *p, int CoutP) {
  const int n = stream_chunk_images(p, CoutP);
  return stream_ctl_bytes(p, n) + ganet_align_up((size_t)n * stream_img_bytes(p, CoutP), 256);
}

template <typename TA, typename TO>
static int stream_launch(const StreamArgs &a, int Cp, int grid, hipStream_t s) {
  if (Cp == 256)
    hipLaunchKernelGGL((lfa_stream_kernel<TA, TO, 256>), dim3(grid), dim3(FNT), 0, s, a);
  else if (Cp == 128)
    hipLaunchKernelGGL((lfa_stream_kernel<TA, TO, 128>), dim3(grid), dim3(FNT), 0, s, a);
  else
    hipLaunchKernelGGL((lfa_stream_kernel<TA, TO, 64>), dim3(grid), dim3(FNT), 0, s, a);
  return ganet_check_launch();
}

// xh: NHWC map in 16-bit arithmetic dtype `ta` with Cp channels; wq: packed weights
// [T][CoutP][Cp]; ws: stream_workspace_bytes() of scratch (control block first).
int stream_forward(const void *xh, const float *offset, const float *mask, const void *wq,
                   const void *bias, void *out, const ganet_dcn_params *p, int ta, int Cp,
                   int CoutP, void *ws, hipStream_t s) {
  const int T = p->kH * p->kW;
  const int HW = p->H * p->W;
  const int chunk = stream_chunk_images(p, CoutP);
  const size_t ctl_bytes = stream_ctl_bytes(p, chunk);
  const size_t out_es = p->dtype == GANET_F32 ? 4 : 2;
  StreamArgs a;
  a.wq = wq;
  a.ctl = static_cast<unsigned int *>(ws);
  a.y = static_cast<unsigned char *>(ws) + ctl_bytes;
  a.bias = bias;
  a.H = p->H;
  a.W = p->W;
  a.Cp = Cp;
  a.Cout = p->Cout;
  a.CoutP = CoutP;
  a.T = T;
  a.kW = p->kW;
  a.pH = p->pH;
  a.pW = p->pW;
  a.dH = p->dH;
  a.dW = p->dW;
  a.out_nhwc = p->out_layout == GANET_NHWC;
  a.relu = p->fuse_relu;
  a.points = p->offset_is_points;
  a.nsplit = stream_nsplit(T, CoutP, Cp);
  if (a.nsplit == 0) return GANET_ERR_UNSUPPORTED;
  a.ncols = T * CoutP / a.nsplit;
  a.pipi = (HW + PITEM - 1) / PITEM;
  a.cipi = (HW + CITEM - 1) / CITEM;
  a.nbands = (HW + BANDPX - 1) / BANDPX;
  a.prod_waves = STREAM_PROD_WAVES;
  a.whole_image = env_int("GANET_STREAM_WHOLE_IMAGE", 0);
  a.debug = env_int("GANET_STREAM_DEBUG_BITS", 0);
  // one workgroup per CU
  int dev = 0, cus = 256;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
    return GANET_ERR_HIP;
  for (int b0 = 0; b0 < p->B; b0 += chunk) {
    const int nb = p->B - b0 < chunk ? p->B - b0 : chunk;
    a.B = nb;
    a.x = static_cast<const unsigned char *>(xh) + (size_t)b0 * HW * Cp * 2;
    a.offset = offset + (size_t)b0 * 2 * T * HW;
    a.mask = mask ? mask + (size_t)b0 * T * HW : nullptr;
    a.out = static_cast<unsigned char *>(out) + (size_t)b0 * HW * p->Cout * out_es;
    // the status word (CTL_ERROR, word 0) is cleared once per call: a later chunk must not
    // erase the failure of an earlier one (its workgroups see the word and stop at once)
    const size_t keep = b0 == 0 ? 0 : 4;
    if (hipMemsetAsync(static_cast<unsigned char *>(ws) + keep, 0, ctl_bytes - keep, s) != hipSuccess)
      return GANET_ERR_HIP;
    int rc;
    if (ta == GANET_BF16 && p->dtype == GANET_F32)
      rc = stream_launch<bf16_t, float>(a, Cp, cus, s);
    else if (ta == GANET_BF16)
      rc = stream_launch<bf16_t, bf16_t>(a, Cp, cus, s);
    else
      rc = stream_launch<f16_t, f16_t>(a, Cp, cus, s);
    if (rc != GANET_OK) return rc;
  }
  return GANET_OK;
}

}  // namespace ganet_dcn
